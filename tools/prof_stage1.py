import sys, ctypes as C
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd.lru import LRURec, init_lru_state_dict
from llamarec_amd.synth import WORKLOADS, synth_users
from llamarec_amd import _lib
lib = _lib.lib()
EXCL = "--no-exclude" not in sys.argv
ONLY = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
for name, U in (("beauty", 22332), ("synth-1m", 4096), ("ml-100k", 610)):
    if ONLY and name not in ONLY:
        continue
    w = WORKLOADS[name]
    hist, labels, n, T = synth_users(name, U)
    model = LRURec.from_state_dict(init_lru_state_dict(w["V"], seed=42))
    ids = torch.from_numpy(hist).cuda()
    model.retrieve_topk(ids, 50, EXCL); torch.cuda.synchronize()
    lib.lr_profile_start(64)
    for _ in range(3): model.retrieve_topk(ids, 50, EXCL)
    torch.cuda.synchronize(); lib.lr_profile_stop()
    for kind, nm in ((4, "encode"), (5, "item_topk")):
        ms, wk, cnt = C.c_double(), C.c_double(), C.c_int64()
        lib.lr_profile_collect(kind, C.byref(ms), C.byref(wk), C.byref(cnt))
        print(name, nm, "avg ms", ms.value / max(1, cnt.value))
