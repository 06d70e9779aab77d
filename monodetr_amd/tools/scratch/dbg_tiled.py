import sys, torch, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from test_msda_gpu import pyramid_problem, KITTI, dev, run_bwd, oracle_bwd
from oracle import msda_oracle
from monodetr_amd import msda_ext
p = pyramid_problem(2, KITTI, "local", seed=4*7+5)
d = dev(p)
gv, gl, ga = run_bwd(msda_ext, d)
rv, rl, ra = msda_oracle.backward(p["value"].double(), p["shapes"], p["level_start"], p["loc"].double(), p["attn"].double(), p["grad_out"].double())
dl = (gl.cpu().double()-rl).abs().amax(-1)   # [B,Lq,M,L,P]
print("gv err", (gv.cpu().double()-rv).abs().max().item(), "gl err", dl.max().item(), "ga err", (ga.cpu().double()-ra).abs().max().item())
bad = (dl > 1e-3).nonzero()
print(len(bad), bad[:10])
print("per level count", [(bad[:,3]==l).sum().item() for l in range(4)], "per b", [(bad[:,0]==b).sum().item() for b in range(2)])
qs = bad[:,1]
print("q range", qs.min().item(), qs.max().item())
i = bad[0]
print(gl.cpu()[tuple(i)], rl[tuple(i)], p["loc"][tuple(i)])
