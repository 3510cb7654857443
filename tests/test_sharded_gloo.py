"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: shard ranges, global-env-id RNG keying and
the rollout-boundary all-gather.  The rank-local stepper is a CPU stand-in built on the TEST-ONLY
oracle (tests may use it); on GPUs the same ShardedRaceEnv wraps the HIP product over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


class CpuStandInEnv:
    """Device-API look-alike (reset_device / step_device / rollout_device) on the CPU oracle."""

    def __init__(self, n, env_id_base, variant=1, seed=5):
        import parity as P
        from oracle_adapter import OracleAdapter

        self.a = OracleAdapter(variant, n, P.tracks()["square"], gates_ahead=1, seed=seed, env_id_base=env_id_base)
        self.num_envs = n

    def reset_device(self):
        return torch.from_numpy(self.a.reset())

    def step_device(self, actions):
        o, r, d, t = self.a.step(actions.numpy())
        return torch.from_numpy(o), torch.from_numpy(r), torch.from_numpy(d.astype(np.uint8)), torch.from_numpy(t.astype(np.uint8))

    def rollout_device(self, actions):
        outs = [self.step_device(actions[k]) for k in range(actions.shape[0])]
        return tuple(torch.stack([o[j] for o in outs]) for j in range(4))



def _free_port():
    """a TCP port the kernel just handed out (a pid-derived port collided with a socket in TIME_WAIT once)"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, n_global, K, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from optimal_quad_control_rl_amd.sharded import ShardedRaceEnv, shard_range

    env = ShardedRaceEnv(n_global, lambda n, base: CpuStandInEnv(n, base))
    assert (env.lo, env.hi) == shard_range(n_global, rank, world)
    env.reset()
    g = torch.Generator().manual_seed(0)
    acts_global = torch.rand((K, n_global, 4), generator=g) * 2 - 1  # same on every rank
    obs, rew, done, trunc = env.rollout(acts_global[:, env.lo:env.hi].contiguous())
    g = env.gather_rollout(obs, rew, done.to(torch.uint8))
    assert g.obs.shape == (world, K, env.num_envs, obs.shape[-1]) and g.done.dtype == torch.uint8
    full_obs, full_rew, full_done = g.global_view()       # strided views [K][world][n]: no second copy of the gathered rollout
    assert full_obs.shape == (K, world, env.num_envs, obs.shape[-1]) and full_done.dtype == torch.bool
    assert full_obs.data_ptr() == g.obs.data_ptr()
    # own shard is found at its global position
    assert torch.equal(full_obs[:, rank], obs) and torch.equal(full_rew[:, rank], rew)
    # the receive buffers are reused: a second gather lands in the same memory
    g2 = env.gather_rollout(obs, rew, done.to(torch.uint8))
    assert g2.obs.data_ptr() == g.obs.data_ptr() and g2.done.data_ptr() == g.done.data_ptr()
    if rank == 0:
        # the gathered rollout equals ONE unsharded env over all global ids (same global-id keyed reset stream)
        ref = CpuStandInEnv(n_global, 0)
        ref.reset_device()
        r_obs, r_rew, r_done, _ = ref.rollout_device(acts_global)
        n = env.num_envs
        assert torch.equal(full_obs.reshape(K, n_global, -1), r_obs) and torch.equal(full_rew.reshape(K, n_global), r_rew)
        assert torch.equal(full_done.reshape(K, n_global), r_done.bool())
        assert r_done.sum() > 0
        # flat row view: row_index(k, g) addresses (step k, global env g)
        rows_obs, rows_rew, rows_done = g.rows()
        for k, ge in ((0, 0), (K - 1, n_global - 1), (7, n + 3)):
            assert torch.equal(rows_obs[g.row_index(k, ge)], r_obs[k, ge]) and rows_rew[g.row_index(k, ge)] == r_rew[k, ge]
            o1, r1, d1 = g.step_of_env(k, ge)
            assert torch.equal(o1, r_obs[k, ge]) and bool(d1) == bool(r_done[k, ge])
        open(os.path.join(tmp, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_world2_sharded_rollout_and_allgather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 64, 150, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _worker_overlap(rank, world, port, n_global, K, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from optimal_quad_control_rl_amd.sharded import ShardedRaceEnv

    g = torch.Generator().manual_seed(1)
    batches = [torch.rand((K, n_global, 4), generator=g) * 2 - 1 for _ in range(5)]   # same on every rank
    env = ShardedRaceEnv(n_global, lambda n, base: CpuStandInEnv(n, base))
    env.reset()
    got = []
    for gr in env.rollouts_overlapped(b[:, env.lo:env.hi].contiguous() for b in batches):
        o, r, d = gr.global_view()
        got.append((o.reshape(K, n_global, -1).clone(), r.reshape(K, n_global).clone(), d.reshape(K, n_global).clone()))
    assert len(got) == len(batches)
    # blocking reference: same env construction, gather_rollout after every rollout
    ref_env = ShardedRaceEnv(n_global, lambda n, base: CpuStandInEnv(n, base))
    ref_env.reset()
    for b, (o, r, d) in zip(batches, got):
        obs, rew, done, _ = ref_env.rollout(b[:, ref_env.lo:ref_env.hi].contiguous())
        fo, fr, fd = ref_env.gather_rollout(obs, rew, done.to(torch.uint8)).global_view()
        assert torch.equal(fo.reshape(K, n_global, -1), o) and torch.equal(fr.reshape(K, n_global), r) and torch.equal(fd.reshape(K, n_global), d)
    if rank == 0:
        open(os.path.join(tmp, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gather_overlapped_with_the_next_collect(tmp_path):
    """the rollout-boundary all-gather started asynchronously into alternating receive buffers, so that the
    gather of rollout i overlaps the collect of rollout i + 1 -- same gathered rollouts as the blocking form, five rollouts in a row."""
    port = _free_port()
    mp.spawn(_worker_overlap, args=(2, port, 32, 40, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_range_partition():
    from optimal_quad_control_rl_amd.sharded import shard_range

    for n, w in ((262144, 8), (65536, 4), (10, 3), (7, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _ddp_worker(rank, world, port, tmp):
    """The collective of data-parallel PPO: ranks that start from the same parameters and average their gradients take the
    same Adam step (here with torch.optim.Adam on CPU standing in for qr_ppo_apply, which needs the GPU)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater, average_across_ranks

    assert MfmaPpoUpdater.data_parallel()          # initialised, world_size 2
    torch.manual_seed(0)
    pol = ActorCritic(17, 4)                        # same initial parameters on both ranks
    opt = torch.optim.Adam(pol.parameters(), lr=3e-4, eps=1e-5)
    g = torch.Generator().manual_seed(100 + rank)   # different data per rank
    obs = torch.randn((256, 17), generator=g)
    loss = pol.pi(obs).pow(2).mean() + pol.value(obs).pow(2).mean()
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in pol.parameters() if p.grad is not None])
    local = flat.clone()
    avg = average_across_ranks(flat)
    others = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(others, local)
    assert torch.allclose(avg, torch.stack(others).mean(0), atol=1e-7)
    off = 0
    for p in pol.parameters():
        if p.grad is not None:
            p.grad.copy_(avg[off:off + p.numel()].view_as(p))
            off += p.numel()
    opt.step()
    after = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    both = [torch.empty_like(after) for _ in range(world)]
    dist.all_gather(both, after)
    assert torch.equal(both[0], both[1])           # identical parameters on every rank after the step
    if rank == 0:
        open(os.path.join(tmp, "ddp_ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gradient_averaging_keeps_ranks_identical(tmp_path):
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ddp_ok").exists()


def _kl_worker(rank, world, port, tmp):
    """Data-parallel target-KL early stop: the minibatch statistics ride behind the gradient in ONE [n + 4] vector
    (qr_ppo_grad), the all-reduce averages them with it, and qr_ppo_apply decides on the averaged KL sum -- so every rank
    takes the same decision even when the rank-local KLs straddle the threshold (a rank-local decision would desynchronise
    the per-minibatch all-reduce)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from optimal_quad_control_rl_amd.ppo import average_across_ranks

    n, B, target_kl = 1000, 2048, 0.02
    g = torch.Generator().manual_seed(7 + rank)
    vec = torch.randn(n + 4, generator=g)
    local_kl_mean = 0.01 if rank == 0 else 0.06            # rank 0 below 1.5 * target_kl = 0.03, rank 1 above
    vec[n + 2] = local_kl_mean * B                          # the KL SUM of this rank's rows
    local_decision = bool(vec[n + 2] > 1.5 * target_kl * B)
    avg = average_across_ranks(vec.clone())
    decision = bool(avg[n + 2] > 1.5 * target_kl * B)       # what ppo_apply_kernel evaluates (kl_limit = 1.5 target_kl B)
    gathered = [torch.empty(1) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor([float(decision)]))
    local = [torch.empty(1) for _ in range(world)]
    dist.all_gather(local, torch.tensor([float(local_decision)]))
    assert gathered[0] == gathered[1]                       # collective decision: identical everywhere
    assert local[0] != local[1]                             # the rank-local rule would have split the ranks
    assert decision == ((0.01 + 0.06) / 2 > 1.5 * target_kl)
    both = [torch.empty_like(avg) for _ in range(world)]
    dist.all_gather(both, avg)
    assert torch.equal(both[0], both[1])                    # gradient and statistics identical after the all-reduce
    if rank == 0:
        open(os.path.join(tmp, "kl_ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_world2_target_kl_decision_is_collective(tmp_path):
    port = _free_port()
    mp.spawn(_kl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "kl_ok").exists()
