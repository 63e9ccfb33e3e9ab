"""CPU oracle for the ws-unet UNet pixel-prediction hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``ws_unet_amd``)
imports from here; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the timed
CPU baseline -- never as the thing shipped.

Parity status: PINNED.  The reference (pure Python on PyTorch, no tests and no
golden vectors of its own -- SURVEY.md section 4) imports in the build container, so
``tests/golden/make_golden.py`` ran the reference's own ``UNet`` / losses /
meters / fabrika on deterministic formula weights and inputs and committed the
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
function here against those vectors.  What is NOT pinned: results on the
published pretrained weights (checkpoints absent from the reference tree,
SURVEY.md F6) -- "parity on published weights unpinned".

Modules
-------
unet_ref      torch-CPU (ATen, the library the reference itself dispatches to)
              restatement of src/unet/model/unet.py forward; autograd gives
              the backward oracle.
np_ops        independent plain-numpy restatement of every operator (conv,
              pool, transposed conv, dropout, loss, WS statistic, AdamW).
evaluate_ref  infere_single / predict_unet / transform restatement.
losses_ref    L1Loss / WSLoss / L1WSLoss.
metrics_ref   AverageMeter / MAEMeter / WSMeter.
"""
