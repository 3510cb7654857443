"""One process for rocprofv3 --kernel-trace --stats: 200 reference-precision minibatch updates at the reference recipe's batch size (5000 rows)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater
dev = torch.device("cuda", 0); L = 24; B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000; R = 100000
pol = ActorCritic(L, 4).to(dev)
obs = torch.randn(R, L, device=dev); act = torch.randn(R, 4, device=dev) * 0.5
old = torch.randn(R, device=dev) * 0.1 - 3; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
perm = torch.randperm(R, device=dev).to(torch.int32)
up = MfmaPpoUpdater(pol, L, dev, B, precision="f32", flags=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for k in range(200):
    s = (k % (R // B)) * B
    up.minibatch(obs, act, old, adv, ret, perm[s:s + B], 3e-4)
torch.cuda.synchronize()
