"""MLP_TRANSPORT=rccl with a world of one, outside pytest, with RCCL's own log: what ncclCommInitRank does on this box."""
import os, sys
os.environ.setdefault("NCCL_DEBUG", "INFO")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("MLP_IMPORT_TORCH"):
    import torch  # noqa: F401
import minilp_amd as M
from minilp_amd import api, dist as md, lpgen
lp = lpgen.gen_sparse_lp(300, 280, 12, 9)
prob = lpgen.build_problem(M.Problem, lp)
s = prob.solve(budget=0, trace=True)
box = md.create_mailbox(1)
print("unique id ...", flush=True)
uid = api.rccl_unique_id()
print("enable_sharding_ex ...", flush=True)
s.enable_sharding_ex(0, 1, box, "rccl", uid)
print("transport:", s.transport(), flush=True)
s.continue_solve(-1)
print("objective", s.objective())
