"""One rank of tests/test_gpu_dp.py::test_two_rank_sharded_evaluate (a fresh process per rank:
`python eval_worker.py RANK WORLD PORT DATASET OUT.csv [MODE [TRIP_RANK]]`).  Two of these share the box's single GPU over gloo and run the REAL
`evaluate.predict_unet_sharded` (BASELINE.json configs[3]: rows split contiguously over the ranks, results all-gathered, every rank
returns the full table in fabrika order) in the default mode."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))


def main():
    rank, world, port, dataset, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    mode = sys.argv[6] if len(sys.argv) > 6 and sys.argv[6] != "default" else None
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "RANK": str(rank), "WORLD_SIZE": str(world)})
    import torch
    from ws_unet_amd import evaluate, parallel
    from gpu_util import gpu_model
    parallel.init_from_env("gloo")
    trip = int(sys.argv[7]) if len(sys.argv) > 7 else -1
    model = gpu_model(2, "he", mode, drop_rate=0.)
    if rank == trip:
        # only THIS rank's shard "overflows": its range flag is set before its first forward, so its model's own first-forward look
        # switches it to 'bf16x3s' mid-pass while the other rank stays planar -- the end-of-pass collective look must bring everybody along
        model._range_flag_tensor(torch.device("cuda")).fill_(1)
    cov = evaluate.predict_unet_sharded(dataset, model, batch_size=2)
    st = evaluate.predict_unet_sharded(dataset, model, stego_method="LSBR", batch_size=2)
    if trip >= 0:
        assert model.mode == "bf16x3s", model.mode
    cov.to_csv(out + ".cover.csv", index=False)
    st.to_csv(out + ".stego.csv", index=False)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
