import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from llamarec_amd.lru import LRURec, init_lru_state_dict
from llamarec_amd.synth import WORKLOADS, synth_users
name, U = sys.argv[1], int(sys.argv[2])
w = WORKLOADS[name]
hist, labels, n, T = synth_users(name, U)
model = LRURec.from_state_dict(init_lru_state_dict(w["V"], seed=42))
ids = torch.from_numpy(hist).cuda()
for _ in range(2): model.retrieve_topk(ids, 50, True)
torch.cuda.synchronize()
