"""GPU time of one evaluate chunk (u8 upload -> /255 -> forward -> WS statistics -> rows) against the bare forward, HIP events on the
current stream; and the wall time per chunk of a queue of chunks submitted back to back (nothing waits) -- where the evaluate loop's
steady state (tools/bench_evaluate.py) loses against bench.py's predict step.      python tools/time_evaluate_chunk.py [batch]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_unet_amd import evaluate, formula, ops
from ws_unet_amd.model import get_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=0.)
m.load_state_dict({k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "he").items()})
m = m.to("cuda")
u8 = torch.from_numpy(formula.synthetic_images(B, 512, 512, seed=5))
pin = u8.pin_memory()
xd = pin.to("cuda")
x01 = ops.u8_to_unit(xd)[:, None]


def ev_time(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def fwd():
    with torch.no_grad():
        return m(x01)


def chunk_device():
    return evaluate.predict_u8_batch(xd, m)


def chunk_with_upload():
    x = pin.to("cuda", non_blocking=True)
    b, l = evaluate.predict_u8_batch(x, m)
    return torch.stack([b, l], dim=1)


print(f"batch {B}: forward {ev_time(fwd):.3f} ms; /255 + forward + statistics {ev_time(chunk_device):.3f} ms; with the upload from pinned memory {ev_time(chunk_with_upload):.3f} ms (GPU time per chunk, queued back to back)")
# the pipeline's pattern: submit k+1, then read the rows of k
torch.cuda.synchronize(); t0 = time.perf_counter(); pend = None; n = 40
for k in range(n):
    h = chunk_with_upload()
    if pend is not None:
        pend.cpu()
    pend = h
pend.cpu(); torch.cuda.synchronize()
print(f"submit(k+1) before rows(k).cpu(): {(time.perf_counter() - t0) / n * 1e3:.3f} ms wall per chunk")
