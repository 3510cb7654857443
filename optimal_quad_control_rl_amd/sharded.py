"""Multi-GPU sharding of the race environment: one process per GPU, independent env shards, and the
RCCL all-gather at the rollout boundary (BASELINE config 4).

Env instances never interact (no cross-env term in R:501-595), so rank r simply simulates global envs
[r*n_local, (r+1)*n_local): its in-kernel Philox stream is keyed by the GLOBAL env id (`env_id_base`), which
makes the union of the shards bit-identical to one big env (tests/test_gpu_scale.py).  The only exchange is
`gather_rollout`, issued once per rollout -- never inside the step kernel's path.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU), so the direct all-gather is per-link bound: one large message per peer per rollout.

What is exchanged and how (round 3): the rollout buffers the kernels wrote -- obs [K][n][L] f32, reward [K][n] f32,
done [K][n] u8 -- go out AS THEY ARE, one `all_gather_into_tensor` each, into receive buffers [world][K][n][...] that
are allocated once and reused.  No packing pass, no float copy of `done` (a byte side channel: K n bytes instead of
4 K n), no permute().reshape() copy of the gathered tensor: peak memory of a gather = the receive buffers themselves.
Consumers index the rank-major result -- `GatheredRollout.rows()` (every (rank, step, env) row, which is all PPO's
minibatch sampling needs) or `global_view()` ([K][world][n] strided views in global env order) -- instead of
materialising a [K][N_global] copy.
"""
import torch
import torch.distributed as dist


def shard_range(num_envs_global, rank, world):
    """Global env index range [lo, hi) owned by `rank` (contiguous, remainder spread over the first ranks)."""
    base, rem = divmod(int(num_envs_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GatheredRollout:
    """Result of RolloutGather.gather(): obs [world][K][n][L] f32, rew [world][K][n] f32, done [world][K][n] u8 -- the
    receive buffers themselves (valid until the next gather of the same shape).  Global env g = rank * n + i."""

    __slots__ = ("obs", "rew", "done")

    def __init__(self, obs, rew, done):
        self.obs, self.rew, self.done = obs, rew, done

    @property
    def world(self):
        return self.obs.shape[0]

    @property
    def steps(self):
        return self.obs.shape[1]

    @property
    def envs_per_rank(self):
        return self.obs.shape[2]

    @property
    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in (self.obs, self.rew, self.done))

    def global_view(self):
        """(obs [K][world][n][L], rew [K][world][n], done [K][world][n] bool) as strided VIEWS: element [k, r, i] is global
        env r * n + i at step k.  (A contiguous [K][N_global] tensor would need a second full-size copy.)"""
        return self.obs.permute(1, 0, 2, 3), self.rew.permute(1, 0, 2), self.done.permute(1, 0, 2).bool()

    def rows(self):
        """(obs [world K n][L], rew [world K n], done [world K n] u8): every sample of the gathered rollout as contiguous
        views -- row index of (step k, global env g) is row_index(k, g)."""
        L = self.obs.shape[-1]
        return self.obs.view(-1, L), self.rew.view(-1), self.done.view(-1)

    def row_index(self, k, g):
        n = self.envs_per_rank
        return ((g // n) * self.steps + k) * n + g % n

    def step_of_env(self, k, g):
        n = self.envs_per_rank
        r, i = divmod(int(g), n)
        return self.obs[r, k, i], self.rew[r, k, i], self.done[r, k, i]


class RolloutGather:
    """The rollout-boundary collective with its receive buffers held across calls (one set per distinct shape)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self._bufs = {}

    def buffers(self, K, n, L, device, obs_dtype=torch.float32, slot=0):
        key = (int(K), int(n), int(L), str(device), obs_dtype, int(slot))
        b = self._bufs.get(key)
        if b is None:
            w = self.world
            b = (torch.empty((w, K, n, L), dtype=obs_dtype, device=device), torch.empty((w, K, n), dtype=torch.float32, device=device),
                 torch.empty((w, K, n), dtype=torch.uint8, device=device))
            self._bufs[key] = b
        return b

    def gather(self, obs, rew, done):
        """obs [K][n][L] f32, rew [K][n] f32, done [K][n] u8 (contiguous, as the rollout kernels write them)."""
        assert obs.is_contiguous() and rew.is_contiguous() and done.is_contiguous()
        assert done.dtype == torch.uint8 and rew.dtype == torch.float32
        K, n, L = obs.shape
        g_obs, g_rew, g_done = self.buffers(K, n, L, obs.device, obs.dtype)
        # rank-major concatenation along dim 0 = the [world] axis of the receive buffers: nothing to rearrange afterwards
        dist.all_gather_into_tensor(g_obs.view(self.world * K, n, L), obs, group=self.group)
        dist.all_gather_into_tensor(g_rew.view(self.world * K, n), rew, group=self.group)
        dist.all_gather_into_tensor(g_done.view(self.world * K, n), done, group=self.group)
        return GatheredRollout(g_obs, g_rew, g_done)

    def gather_async(self, obs, rew, done, slot=0):
        """The same three collectives started WITHOUT waiting (async_op: RCCL runs them on its own stream, so the next collect kernel on the
        caller's stream overlaps them -- round 5's review: the blocking form leaves ~22 us per step-equivalent of xGMI time on the table).
        Returns a PendingGather; `.wait()` gives the GatheredRollout.  `slot` (0 / 1) selects one of two sets of receive buffers: the
        gather of rollout i may still be in flight while rollout i + 1 is collected and gathered into the other set.  The SEND buffers
        (obs, rew, done) must not be overwritten before `.wait()` -- alternate them too (ShardedRaceEnv.rollouts_overlapped does)."""
        assert obs.is_contiguous() and rew.is_contiguous() and done.is_contiguous()
        assert done.dtype == torch.uint8 and rew.dtype == torch.float32
        K, n, L = obs.shape
        g_obs, g_rew, g_done = self.buffers(K, n, L, obs.device, obs.dtype, slot)
        # (on RCCL the collective's stream first waits for the caller's current stream: it sees the collect kernel's writes)
        works = [dist.all_gather_into_tensor(g_obs.view(self.world * K, n, L), obs, group=self.group, async_op=True),
                 dist.all_gather_into_tensor(g_rew.view(self.world * K, n), rew, group=self.group, async_op=True),
                 dist.all_gather_into_tensor(g_done.view(self.world * K, n), done, group=self.group, async_op=True)]
        return PendingGather(works, GatheredRollout(g_obs, g_rew, g_done))


class PendingGather:
    def __init__(self, works, result):
        self._works, self._result = works, result

    def wait(self):
        for w in self._works:
            w.wait()   # (on RCCL: makes the caller's current stream wait for the collective; no host synchronisation)
        self._works = []
        return self._result


class ShardedRaceEnv:
    """Rank-local shard of a `num_envs_global`-env race environment.

    `env_factory(n_local, env_id_base)` builds the local env (default: the HIP product); tests inject a CPU
    stand-in to exercise the partition / gather logic over gloo.
    """

    def __init__(self, num_envs_global, env_factory, rank=None, world=None, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.num_envs_global = int(num_envs_global)
        self.lo, self.hi = shard_range(num_envs_global, self.rank, self.world)
        self.num_envs = self.hi - self.lo
        if num_envs_global % self.world:
            raise ValueError("all_gather_into_tensor needs equal shards: num_envs_global % world_size != 0")
        self.env = env_factory(self.num_envs, self.lo)
        self._gather = RolloutGather(group)

    def reset(self):
        return self.env.reset_device()

    def step(self, actions_local):
        return self.env.step_device(actions_local)

    def rollout(self, actions_local):
        return self.env.rollout_device(actions_local)

    def gather_rollout(self, obs, rew, done):
        """All ranks receive the full rollout as a GatheredRollout (rank-major receive buffers, reused between calls)."""
        return self._gather.gather(obs, rew, done)

    def rollouts_overlapped(self, action_batches):
        """Generator over `action_batches` (local actions [K][n][4] per rollout): yields the GatheredRollout of rollout i while rollout
        i + 1 is already being collected -- the gather of one rollout overlaps the collect kernel of the next (two alternating sets of
        send and receive buffers).  A yielded GatheredRollout stays valid until the rollout after the next one has been gathered."""
        pending, out_bufs = None, [None, None]
        for i, acts in enumerate(action_batches):
            slot = i & 1
            out = self.env.rollout_device(acts, out_bufs[slot]) if out_bufs[slot] is not None and self._reuses_out() else self.env.rollout_device(acts)
            out_bufs[slot] = out
            obs, rew, done = out[0], out[1], out[2]
            nxt = self._gather.gather_async(obs.contiguous(), rew.contiguous(), done.to(torch.uint8).contiguous(), slot)
            if pending is not None:
                yield pending.wait()
            pending = nxt
        if pending is not None:
            yield pending.wait()

    def _reuses_out(self):
        import inspect
        try:
            return "out" in inspect.signature(self.env.rollout_device).parameters
        except (TypeError, ValueError):
            return False
