#!/usr/bin/env python3
"""Time one PPO minibatch update on the MI355X: the matrix-core kernels (qr_ppo_minibatch) vs the torch autograd +
torch.optim.Adam update they replace, on the same synthetic rollout rows.  Prints one JSON line.

    python tools/bench_ppo_update.py [--obs-len 17] [--minibatch 16384] [--iters 200]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater


def measure(L=17, B=16384, R=65536 * 8, iters=100, with_torch=True):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    obs = torch.randn((R, L), device=dev); act = torch.randn((R, 4), device=dev) * 0.5
    old_lp = torch.randn(R, device=dev) * 0.1 - 3.0; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
    perm = (torch.arange(R, device=dev) if os.environ.get("QR_BENCH_SEQUENTIAL_ROWS") else torch.randperm(R, device=dev)).to(torch.int32)

    def timed(fn, n):
        for k in range(min(10, n)):
            fn(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            fn(k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    pol = ActorCritic(L, 4).to(dev)
    up = MfmaPpoUpdater(pol, L, dev, B)
    nb = R // B

    def native_step(k):   # the training loop's calling pattern: one adv-statistics launch per epoch, three launches per minibatch
        if k % nb == 0:
            up.begin_epoch(adv, perm, B)
        up.minibatch(obs, act, old_lp, adv, ret, perm[(k % nb) * B:(k % nb + 1) * B], 3e-4)

    t_native = timed(native_step, iters)

    # the same updates as ONE replayed graph per epoch (qr_ppo_epoch: what PPO.train() calls), nb minibatches per epoch
    perm_dev = perm[:nb * B].clone()

    def epoch_step(k):
        up.epoch(obs, act, old_lp, adv, ret, perm_dev, B, 3e-4, device_shuffle=True)

    t_epoch = timed(epoch_step, max(3, iters // nb)) / nb
    flops = 2 * B * 3 * 2 * (L * 120 + 2 * 120 * 120 + 2.5 * 120)   # 2 nets x (fwd + 2 bwd GEMMs) x 2 flop/MAC, useful MACs only
    out = {"what": "one PPO minibatch update (both 3x120 networks: forward, loss, backward, grad-norm clip, Adam): qr_ppo_minibatch "
                   "vs torch autograd + torch.optim.Adam on the same rows", "obs_len": L, "minibatch": B,
           "native_us": t_native * 1e6, "native_samples_per_s": B / t_native, "useful_TFLOPs": flops / t_native / 1e12,
           "epoch_us": t_epoch * 1e6, "epoch_minibatches": nb, "epoch_useful_TFLOPs": flops / t_epoch / 1e12}
    st = up.status()   # (stopped, optimiser steps, skipped non-finite, barrier timeouts): the last two must be 0
    out.update(status_skipped_nonfinite=st[2], status_barrier_timeouts=st[3])
    up.close()
    # the reference-precision update (precision="f32": qr_ppo_grad_f32class, three bf16 pieces per GEMM operand, + qr_ppo_apply), the A/B path:
    # the same rows, and the reference recipe's own minibatch of 5 000 rows (R:785-792)
    try:
        for rows_f32, key in ((B, "f32class_us"), (5000, "f32class_us_5000_rows")):
            up32 = MfmaPpoUpdater(pol, L, dev, rows_f32, precision="f32")
            nb32 = min(R // rows_f32, 32)
            t32 = timed(lambda k: up32.minibatch(obs, act, old_lp, adv, ret, perm[(k % nb32) * rows_f32:(k % nb32 + 1) * rows_f32], 3e-4), max(40, iters // 2))
            out[key] = t32 * 1e6
            up32.close()
    except Exception as ex:  # pragma: no cover
        out["f32class_error"] = repr(ex)
    if with_torch:
        ref = ActorCritic(L, 4).to(dev)
        opt = torch.optim.Adam(ref.parameters(), lr=3e-4, eps=1e-5)

        def torch_step(k):
            idx = perm[(k % nb) * B:(k % nb + 1) * B].long()
            x = adv[idx]; x = (x - x.mean()) / (x.std() + 1e-8)
            lp, ent = ref.log_prob_entropy(obs[idx], act[idx])
            ratio = (lp - old_lp[idx]).exp()
            loss = -torch.min(x * ratio, x * ratio.clamp(0.8, 1.2)).mean() + 0.5 * torch.nn.functional.mse_loss(ref.value(obs[idx]), ret[idx])
            opt.zero_grad(set_to_none=True); loss.backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5); opt.step()

        t_torch = timed(torch_step, max(20, iters // 4))
        out.update(torch_us=t_torch * 1e6, speedup=t_torch / t_native)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--obs-len", type=int, default=17)
    ap.add_argument("--minibatch", type=int, default=16384)
    ap.add_argument("--rows", type=int, default=65536 * 32)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    print(json.dumps(measure(a.obs_len, a.minibatch, a.rows, a.iters)))
