"""Time one PPO minibatch update (gradient + clip / Adam) at 5 000 ... 131 072 rows: the reference-precision path (precision="f32":
qr_ppo_grad_f32class + qr_ppo_apply) next to the f16-operand kernels (qr_ppo_minibatch), stream launches as a training loop issues them."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater
dev = torch.device("cuda", 0)
L = 24
for B in (5000, 16384, 65536, 131072):
    R = max(4 * B, 65536)
    pol = ActorCritic(L, 4).to(dev)
    obs = torch.randn(R, L, device=dev); act = torch.randn(R, 4, device=dev) * 0.5
    old = torch.randn(R, device=dev) * 0.1 - 3; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
    perm = torch.randperm(R, device=dev).to(torch.int32)
    for prec in ("f32", "f16-operands"):
        up = MfmaPpoUpdater(pol, L, dev, B, precision=prec)
        for k in range(3): up.minibatch(obs, act, old, adv, ret, perm[(k % 3) * B:(k % 3 + 1) * B].contiguous(), 3e-4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for k in range(n): up.minibatch(obs, act, old, adv, ret, perm[(k % 3) * B:(k % 3 + 1) * B].contiguous(), 3e-4)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
        print("B=%6d %-13s %8.1f us per minibatch update (%.2f G rows/s)" % (B, prec, t * 1e6, B / t / 1e9))
        up.close()
