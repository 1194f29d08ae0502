"""Test doubles (tests only): numpy/oracle stand-ins for the three HIP hooks of
haconvdr_amd.sharded.ShardedSearcher, so the distributed plumbing can run on CPU with gloo.
They restate the packed-key format of include/haconvdr.h."""
import numpy as np
import torch

FMAX = np.finfo(np.float32).max


def pack_keys(D, pos):
    """(score float32, position int64; -1 = empty) -> uint64 keys viewed as int64."""
    D = np.ascontiguousarray(D, np.float32) + np.float32(0.0)
    bits = D.view(np.uint32)
    neg = (bits & np.uint32(0x80000000)) != 0
    ordv = np.where(neg, ~bits, bits | np.uint32(0x80000000)).astype(np.uint64)
    keys = (ordv << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - pos.astype(np.uint64) % np.uint64(1 << 32))
    keys[pos < 0] = 0
    return keys.view(np.int64)


def unpack_keys(keys):
    k = np.ascontiguousarray(keys).view(np.uint64)
    ordv = (k >> np.uint64(32)).astype(np.uint32)
    bits = np.where((ordv & np.uint32(0x80000000)) != 0, ordv & np.uint32(0x7FFFFFFF), ~ordv)
    D = bits.view(np.float32).copy()
    pos = (np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64)
    empty = k == 0
    D[empty] = -FMAX
    pos[empty] = -1
    return D, pos


def make_local_keys(oracle, x_shard):
    def local_keys(q, k, base):
        D, I = oracle.flat_ip_search(x_shard, q.numpy(), k)
        return torch.from_numpy(pack_keys(D, np.where(I >= 0, I + base, -1)))
    return local_keys


def merge(lists):
    a = lists.numpy().view(np.uint64)                      # [R, nq, k]
    R, nq, k = a.shape
    out = np.zeros((nq, k), np.uint64)
    for i in range(nq):
        c = np.sort(a[:, i, :].reshape(-1))[::-1]
        c = c[c != 0][:k]
        out[i, :len(c)] = c
    return torch.from_numpy(out.view(np.int64))


def to_results(keys, id_map=None):
    D, pos = unpack_keys(keys.numpy())
    if id_map is not None:
        idm = np.asarray(id_map)
        pos = np.where(pos >= 0, idm[np.clip(pos, 0, None)], -1)
    return D, pos
