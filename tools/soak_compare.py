"""Loss trajectories of tools/soak_train.py runs side by side: python tools/soak_compare.py ref.json a.json b.json ...
Prints, per run, the loss at a few steps and the largest |loss - loss_ref| / loss_ref over windows of the run (the runs start from the same
weights and see the same batch every step, so they differ only by the arithmetic -- and by the chaos of training that amplifies any difference)."""
import json
import sys

if __name__ == "__main__":
    runs = [json.load(open(p)) for p in sys.argv[1:]]
    ref = runs[0]["loss"]
    n = len(ref)
    marks = [0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 250, n - 1]
    marks = [m for m in marks if m < n]
    print("| run | " + " | ".join(f"step {m}" for m in marks) + " |")
    print("|---|" + "---|" * len(marks))
    for r in runs:
        name = f"{r['train_mode']}" + (f" / products {r['train_products']}" if r["train_mode"] == "f16f8p" else "")
        print(f"| {name} | " + " | ".join(f"{r['loss'][m]:.6f}" for m in marks) + " |")
    print()
    print("| run vs " + runs[0]["train_mode"] + " | max rel dev steps 0-9 | 10-49 | 50-149 | 150-end | final loss |")
    print("|---|---|---|---|---|---|")
    for r in runs[1:]:
        l = r["loss"]
        def dev(a, b):
            b = min(b, n)
            return max(abs(l[i] - ref[i]) / ref[i] for i in range(a, b)) if a < b else float("nan")
        name = f"{r['train_mode']}" + (f" / products {r['train_products']}" if r["train_mode"] == "f16f8p" else "")
        print(f"| {name} | {dev(0, 10):.2e} | {dev(10, 50):.2e} | {dev(50, 150):.2e} | {dev(150, n):.2e} | {l[-1]:.6f} |")
