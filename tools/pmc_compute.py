#!/usr/bin/env python3
"""Compute-side PMC evidence for the kernels that are NOT HBM-bound (rollout_kernel, rollout_policy_kernel, the PPO
update kernels) and for the per-step kernel: instruction counts, MFMA busy cycles, wave cycles, issue stalls.

  probe (run under rocprofv3, one pass per counter group; tools/run_pmc_compute.sh does both):
      rocprofv3 --kernel-trace --pmc <group> --output-format csv -d D -o p -- python tools/pmc_compute.py probe
  summary:
      python tools/pmc_compute.py summarise D1/p_counter_collection.csv D2/p_counter_collection.csv [..] out.json

Counter groups (8 SQ slots per pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
  A: SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  B: SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
  C: SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU
     -> f32 flop per env-step = 64 lanes x (ADD + MUL + TRANS + 2 FMA) + 512 x MFMA_MOPS_F32  (+ 512 x MFMA_MOPS_F16 counted separately)
Units (guide, "Per-instruction cycle constants"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_INSTS_* count wave-instructions.

Derived per kernel (all per launch):
  valu_floor_cycles  = SQ_INSTS_VALU_nonMFMA x 2 cycles / (waves resident per SIMD = 1 at N = 65 536) -- the f32 vector issue
                       floor of ONE wave (v_fma_f32 wave64 = 2 cycles on the 32-lane SIMD)
  mfma_busy_frac     = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)
  wave_active_frac   = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES,  stall fractions likewise
"""
import csv
import json
import os
import sys

csv.field_size_limit(1 << 30)
GROUP_A = "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
GROUP_B = "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"
GROUP_C = "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"
GROUP_D = "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_INSTS_VALU"
K_FUSED = 32       # steps per fused launch in the probe
N = int(os.environ.get("QR_PMC_ENVS", "65536"))         # envs of the probe (round 4: also run at 1 Mi envs for the throughput regime)
ONLY_ENV = os.environ.get("QR_PMC_ONLY_ENV", "0") == "1"  # skip the predecessor-env and PPO legs of the probe
# (the fused rollout launches as rollout_stash_kernel<V, GA> up to one workgroup per CU and as rollout_kernel<V, GA> beyond)
KERNELS = ("q3_step_kernel<", "q3_rollout_kernel<", "step_kernel<0", "step_kernel<1", "rollout_stash_kernel<0", "rollout_stash_kernel<1",
           "rollout_fast_mlp_kernel<0", "rollout_fast_kernel<0", "rollout_lean_mlp_kernel<0",
           "rollout_kernel<0", "rollout_kernel<1", "rollout_policy_kernel<0", "rollout_policy_kernel<1", "ppo_")


def probe():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from optimal_quad_control_rl_amd.policy import MfmaPolicy
    from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater

    dev = torch.device("cuda", 0)
    for variant in ("e2e", "indi"):
        env = bench.make_env(variant, N, 1, 0)
        L = env.state_len
        gen = torch.Generator(device=dev).manual_seed(0)
        actions = torch.rand((K_FUSED, N, 4), device=dev, generator=gen) * 2 - 1
        out = (torch.empty((K_FUSED, N, L), device=dev), torch.empty((K_FUSED, N), device=dev),
               torch.empty((K_FUSED, N), dtype=torch.uint8, device=dev), torch.empty((K_FUSED, N), dtype=torch.uint8, device=dev))
        env.reset_device()
        env.step_sequence_device(actions, out)          # K_FUSED step kernels
        for _ in range(3):
            env.rollout_device(actions, out)            # fused rollout kernel
        torch.manual_seed(0)
        net = ActorCritic(L, 4).to(dev)
        pol = MfmaPolicy(L, dev.index).load_torch(net.pi)
        res = None
        for r in range(3):
            res = env.rollout_policy_device(pol, K_FUSED, torch.zeros(4), noise_seed=0, first_step=r * K_FUSED,
                                            out=None if res is None else res[:6])
        torch.cuda.synchronize()
        env.close()
    if ONLY_ENV:
        print("pmc compute probe done (env kernels only)")
        return
    # predecessor envs (include/quad3d.h): hover (f64) and gates (f32)
    from optimal_quad_control_rl_amd.quad3d import Quadcopter3DVec, Quadcopter3DVecGates
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_quad3d import Q3_TRACK
    for kind in ("hover", "gates"):
        env = Quadcopter3DVec(N) if kind == "hover" else Quadcopter3DVecGates(N, *Q3_TRACK)
        acts = torch.rand((K_FUSED, N, 4), device=dev) * 2 - 1
        env.reset_device()
        for k in range(8):
            env.step_device(acts[k])
        for _ in range(3):
            env.rollout_device(acts)
        torch.cuda.synchronize()
        env.close()
    # PPO minibatch update (17-float INDI observation, 16 384-row minibatches)
    L, B, R = 17, 16384, 65536 * 4
    obs = torch.randn((R, L), device=dev); act = torch.randn((R, 4), device=dev) * 0.5
    old_lp = torch.randn(R, device=dev) * 0.1 - 3.0; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
    perm = torch.randperm(R, device=dev).to(torch.int32)
    up = MfmaPpoUpdater(ActorCritic(L, 4).to(dev), L, dev, B)
    for k in range(12):
        up.minibatch(obs, act, old_lp, adv, ret, perm[(k % 16) * B:(k % 16 + 1) * B], 3e-4)
    torch.cuda.synchronize()
    print("pmc compute probe done")


def short_name(k):
    for pat in KERNELS:
        if pat in k:
            if pat == "ppo_":
                i = k.index("ppo_")
                j = i
                while j < len(k) and (k[j].isalnum() or k[j] == "_"):
                    j += 1
                return k[i:j]
            i = k.index(pat)
            tail = k[i:]
            name = tail.split(">")[0] + ">"
            return ("qr::" + name) if k[max(0, i - 4):i] == "qr::" else name   # the symbol as rocprofv3 prints it (bench.py's lookup key)
    return None


def summarise(paths, out_path):
    acc = {}
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = short_name(r["Kernel_Name"])
                if name is None:
                    continue
                acc.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    res = {"counters": "rocprofv3 --kernel-trace --pmc (two SQ passes, groups A and B of tools/pmc_compute.py), per launch "
                       "(mean over launches after the first two); N = %d envs, fused kernels: %d steps per launch" % (N, K_FUSED),
           "n_envs": N, "commit": os.environ.get("QR_COMMIT", "unknown"),
           "units": "SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* = quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES = cycles "
                    "summed over SIMDs; SQ_INSTS_* = wave-instructions; GRBM_GUI_ACTIVE = GPU cycles",
           "kernels": {}}
    for name, ctr in sorted(acc.items()):
        m = {c: (sum(v[2:]) / len(v[2:]) if len(v) > 2 else sum(v) / len(v)) for c, v in ctr.items()}
        m["launches"] = max(len(v) for v in ctr.values())
        waves = m.get("SQ_WAVES", 0.0)
        d = {}
        if waves:
            steps = K_FUSED if "rollout" in name else 1
            valu = m.get("SQ_INSTS_VALU", 0.0) - m.get("SQ_INSTS_MFMA", 0.0)
            d["waves"] = waves
            d["valu_insts_per_wave_step"] = valu / waves / steps
            d["mfma_insts_per_wave_step"] = m.get("SQ_INSTS_MFMA", 0.0) / waves / steps
            d["salu_insts_per_wave_step"] = m.get("SQ_INSTS_SALU", 0.0) / waves / steps
            d["lds_insts_per_wave_step"] = m.get("SQ_INSTS_LDS", 0.0) / waves / steps
            d["vmem_insts_per_wave_step"] = m.get("SQ_INSTS_VMEM", 0.0) / waves / steps
            d["wave_cycles_per_wave_step"] = 4.0 * m.get("SQ_WAVE_CYCLES", 0.0) / waves / steps
            d["valu_issue_floor_cycles_per_wave_step"] = 2.0 * valu / waves / steps
            d["mfma_busy_cycles_per_wave_step"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / waves / steps
            wc = m.get("SQ_WAVE_CYCLES", 0.0)
            if wc:
                d["active_inst_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
                d["wait_inst_frac"] = m.get("SQ_WAIT_INST_ANY", 0.0) / wc
                d["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0.0) / wc
                d["active_valu_frac"] = m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
                # fraction of the wave's lifetime spent at the f32 vector issue floor / on the matrix core
                d["valu_floor_frac_of_wave"] = d["valu_issue_floor_cycles_per_wave_step"] / d["wave_cycles_per_wave_step"]
                d["mfma_busy_frac_of_wave"] = d["mfma_busy_cycles_per_wave_step"] / d["wave_cycles_per_wave_step"]
            if "SQ_INSTS_VALU_FMA_F64" in m:
                d["f64_vector_flop_per_env_step"] = 64.0 * (m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0)
                                                            + m.get("SQ_INSTS_VALU_TRANS_F64", 0)
                                                            + 2.0 * m.get("SQ_INSTS_VALU_FMA_F64", 0)) / (waves * 64.0) / steps
            if "SQ_INSTS_VALU_FMA_F32" in m:
                envs = waves * 64.0
                vec = 64.0 * (m.get("SQ_INSTS_VALU_ADD_F32", 0) + m.get("SQ_INSTS_VALU_MUL_F32", 0) + m.get("SQ_INSTS_VALU_TRANS_F32", 0)
                              + 2.0 * m.get("SQ_INSTS_VALU_FMA_F32", 0))
                d["f32_vector_flop_per_env_step"] = vec / envs / steps
                d["f32_mfma_flop_per_env_step"] = 512.0 * m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / envs / steps
                d["f16_mfma_flop_per_env_step"] = 512.0 * m.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) / envs / steps
                d["f32_flop_per_env_step"] = d["f32_vector_flop_per_env_step"] + d["f32_mfma_flop_per_env_step"]
            if m.get("GRBM_GUI_ACTIVE"):
                d["gpu_cycles_per_step"] = m["GRBM_GUI_ACTIVE"] / steps
                d["mfma_busy_frac_of_gpu"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] * 1024.0)
        res["kernels"][name] = {"raw_per_launch": m, "derived": d}
    json.dump(res, open(out_path, "w"), indent=1)
    for name, v in res["kernels"].items():
        print(name, json.dumps({k: round(x, 3) for k, x in v["derived"].items()}))


if __name__ == "__main__":
    if sys.argv[1] == "probe":
        probe()
    elif sys.argv[1] == "groups":
        print(GROUP_A)
        print(GROUP_B)
        print(GROUP_C)
        print(GROUP_D)
    else:
        summarise(sys.argv[2:-1], sys.argv[-1])
