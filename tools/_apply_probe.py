import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater
dev = torch.device("cuda", 0); torch.manual_seed(0)
L, B, R = 17, 16384, 65536 * 8
obs = torch.randn((R, L), device=dev); act = torch.randn((R, 4), device=dev) * 0.5
old_lp = torch.randn(R, device=dev) * 0.1 - 3.0; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
perm = torch.randperm(R, device=dev).to(torch.int32)
pol = ActorCritic(L, 4).to(dev); up = MfmaPpoUpdater(pol, L, dev, B)
for k in range(100):
    up.grad(obs, act, old_lp, adv, ret, perm[(k % 32) * B:(k % 32 + 1) * B])
torch.cuda.synchronize()
