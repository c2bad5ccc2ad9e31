import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import bench
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4_informative
def shard(label):
    r = [bench.post_correct_phase(0, 1, None, reps=3, ns=ns)["seconds"] * 1e3 for ns in (2340, 293)]
    print("%-70s 2340 draws %.2f ms, 293 draws %.2f ms" % (label, r[0], r[1]), flush=True)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
plan = FilterPlan(negbin_m4_informative(1000, seed=103), 65536, method="bsf", keep_history=True, force="grid", n_rep=32, factored_weights=True)
plan.run(5, hip_stream=stream); torch.cuda.synchronize()
plan.close()
shard("FIRST post-correction behind a closed full-output batch (185 GB)")
shard("again")
