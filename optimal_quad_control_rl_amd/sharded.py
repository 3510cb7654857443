"""Multi-GPU sharding of the race environment: one process per GPU, independent env shards, and ONE
RCCL collective at the rollout boundary (BASELINE config 4).

Env instances never interact (no cross-env term in R:501-595), so rank r simply simulates global envs
[r*n_local, (r+1)*n_local): its in-kernel Philox stream is keyed by the GLOBAL env id (`env_id_base`), which
makes the union of the shards bit-identical to one big env (tests/test_gpu_scale.py).  The only exchange is
`gather_rollout`: an all-gather of the packed [obs | reward | done] rollout shard, issued once per rollout --
never inside the step kernel's path.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so this direct
all-gather is per-link bound: one large message per peer per rollout, not one per step.
"""
import torch
import torch.distributed as dist


def shard_range(num_envs_global, rank, world):
    """Global env index range [lo, hi) owned by `rank` (contiguous, remainder spread over the first ranks)."""
    base, rem = divmod(int(num_envs_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_rollout(obs, rew, done):
    """[K,n,L] f32, [K,n] f32, [K,n] u8/bool -> one contiguous [K,n,L+2] f32 message."""
    return torch.cat([obs, rew.unsqueeze(-1), done.to(obs.dtype).unsqueeze(-1)], dim=-1).contiguous()


def unpack_rollout(packed):
    """Inverse of pack_rollout on a gathered [...,L+2] tensor."""
    return packed[..., :-2], packed[..., -2], packed[..., -1] > 0.5


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

    def reset(self):
        return self.env.reset_device()

    def step(self, actions_local):
        return self.env.step_device(actions_local)

    def rollout(self, actions_local):
        return self.env.rollout_device(actions_local)

    def gather_rollout(self, obs, rew, done):
        """All ranks receive the full rollout: obs[K, N_global, L], rew[K, N_global], done[K, N_global]."""
        packed = pack_rollout(obs, rew, done)  # [K, n, L+2]
        K = packed.shape[0]
        flat = torch.empty((self.world * K,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(flat, packed, group=self.group)  # concatenation along dim 0 (rank-major)
        gathered = flat.view((self.world, K) + tuple(packed.shape[1:]))
        # [world, K, n, L+2] -> [K, world*n, L+2]: rank-major = global env order
        full = gathered.permute(1, 0, 2, 3).reshape(packed.shape[0], self.world * packed.shape[1], packed.shape[2])
        return unpack_rollout(full)
