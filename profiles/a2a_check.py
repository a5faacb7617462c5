"""profiles/a2a_check.py -- all_to_all_single of a large uint8 tensor in a one-rank nccl group (MSNV_DIST_FORCE=1 under torchrun): does the receive buffer equal the send buffer?"""
import os, sys, numpy as np
sys.path.insert(0, ".")
from metasnv_amd import parallel
parallel.init_from_env(force=True)
import torch
for mb in (64, 600, 1100, 2300):
    a = np.random.default_rng(mb).integers(0, 255, size=mb << 20, dtype=np.uint8)
    got = parallel.exchange_records([a], keep_on_device=True)
    back = got.tensor.cpu().numpy()
    same = np.array_equal(back, a)
    print(mb, "MB", "equal" if same else "DIFFERENT at byte %d" % int(np.nonzero(back != a)[0][0]), flush=True)
parallel.finalize()
