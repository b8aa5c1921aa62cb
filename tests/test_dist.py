"""CPU: the row-sharded table logic - fake-rank simulation (pure index arithmetic,
bit-exact) and a real 2-process gloo run of lookup / gradient push / dense all-reduce."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from recman_amd import dist as rd


def cpu_gather(table, rows, out):
    out.copy_(table[rows.clamp(min=0)] * (rows >= 0).unsqueeze(1))  # id -1 = empty slot -> zero row


ZOFF = torch.zeros(1, dtype=torch.int64)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_fake_rank_routing_is_a_bit_exact_permutation(world):
    g = torch.Generator().manual_seed(world)
    R, D = 1000 + world, 8
    table = torch.randn(R, D, generator=g)
    shards = [table[r::world] for r in range(world)]
    assert [s.shape[0] for s in shards] == [rd.shard_rows(R, r, world) for r in range(world)]
    rows = torch.randint(0, R, (777,), generator=g)
    rows[:50] = rows[0]  # heavy duplicates
    pos, counts, local = rd.route_torch(rows.view(-1, 1), ZOFF, world)
    assert int(counts.sum()) == 777 and torch.equal(torch.sort(pos).values, torch.arange(777))
    # owner-side gather per bucket; occurrence o then reads row pos[o]
    pieces, off = [], 0
    for w in range(world):
        ids = local[off: off + int(counts[w])]
        members = torch.nonzero((pos >= off) & (pos < off + int(counts[w]))).reshape(-1)
        assert bool(((rows[members] % world) == w).all())
        pieces.append(shards[w][ids])
        off += int(counts[w])
    bucketed = torch.cat(pieces)
    assert torch.equal(bucketed[pos], table[rows])


def test_world_one_table_roundtrip():
    g = torch.Generator().manual_seed(0)
    R, D = 50, 4
    full = torch.randn(R, D, generator=g)
    st = rd.ShardedTable(R, D, 0, 1, "cpu", cpu_gather, rd.route_torch)
    st.load_global(full, bias=torch.arange(R).float(), lin=-torch.arange(R).float())
    rows = torch.randint(0, R, (33,), generator=g)
    buck, ex = st.lookup(rows.view(-1, 1), ZOFF)
    got = buck[ex.pos]
    assert torch.equal(got[:, :D], full[rows])
    assert torch.equal(got[:, D], rows.float()) and torch.equal(got[:, D + 1], -rows.float())
    ids, grows = st.push_grads(ex, buck)
    dense = torch.zeros(R, D + rd.PAD).index_add_(0, ids, grows)
    want = torch.zeros(R, D + rd.PAD).index_add_(0, rows, got)
    assert torch.allclose(dense, want)


def test_shard_checkpoint_roundtrip_and_reshard(tmp_path):
    """Per-rank shard files: same-world restore is exact; a 4-rank checkpoint re-shards onto 3
    and onto 1 rank with every global row in its new home."""
    g = torch.Generator().manual_seed(3)
    R, D = 103, 4
    full = torch.randn(R, D, generator=g)
    bias, lin = torch.randn(R, generator=g), torch.randn(R, generator=g)
    path = str(tmp_path / "table")
    saved = []
    for r in range(4):
        st = rd.ShardedTable(R, D, r, 4, "cpu", cpu_gather, rd.route_torch)
        st.load_global(full, bias, lin)
        st.save(path)
        saved.append(st.shard.clone())
    for r in range(4):
        st = rd.ShardedTable(R, D, r, 4, "cpu", cpu_gather, rd.route_torch)
        st.load(path)
        assert torch.equal(st.shard, saved[r])
    for world in (3, 1):
        for r in range(world):
            st = rd.ShardedTable(R, D, r, world, "cpu", cpu_gather, rd.route_torch)
            st.load(path, saved_world=4)
            want = rd.ShardedTable(R, D, r, world, "cpu", cpu_gather, rd.route_torch)
            want.load_global(full, bias, lin)
            assert torch.equal(st.shard, want.shard)
    other = rd.ShardedTable(R + 1, D, 0, 4, "cpu", cpu_gather, rd.route_torch)
    with pytest.raises(ValueError):
        other.load(path)


def _worker(rank, world, port, R, D, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.randn(R, D, generator=torch.Generator().manual_seed(123))
        bias = torch.arange(R).float()
        st = rd.ShardedTable(R, D, rank, world, "cpu", cpu_gather, rd.route_torch)
        st.load_global(full, bias=bias)
        all_rows = [torch.randint(0, R, (n + 7 * r,), generator=torch.Generator().manual_seed(10 + r))
                    for r in range(world)]
        rows = all_rows[rank]
        buck, ex = st.lookup(rows.view(-1, 1), ZOFF)
        got = buck[ex.pos]
        assert torch.equal(got[:, :D], full[rows]), "lookup mismatch"
        assert torch.equal(got[:, D], bias[rows])
        # gradient push: every rank sends grad = f(its rows); the owner must receive the sum
        grads = [torch.cat([full[rw] * (r + 1), torch.ones(len(rw), rd.PAD)], 1) for r, rw in enumerate(all_rows)]
        bucketed = torch.empty_like(grads[rank])
        bucketed[ex.pos] = grads[rank]
        ids, grows = st.push_grads(ex, bucketed)
        mine = torch.zeros(st.shard.shape[0], D + rd.PAD, dtype=torch.float64).index_add_(0, ids, grows.double())
        want = torch.zeros(R, D + rd.PAD, dtype=torch.float64)
        for rw, gr in zip(all_rows, grads):
            want.index_add_(0, rw, gr.double())
        assert torch.allclose(mine, want[rank::world], atol=1e-9), "gradient push mismatch"
        # dense all-reduce (mean over ranks)
        gd = {"b": torch.full((3,), float(rank + 1)), "a": torch.full((2, 2), float(10 * (rank + 1)))}
        rd.allreduce_dense(gd, world)
        mean = sum(range(1, world + 1)) / world
        assert torch.allclose(gd["b"], torch.full((3,), mean)) and torch.allclose(gd["a"], torch.full((2, 2), 10 * mean))
    finally:
        dist.destroy_process_group()


def test_fixed_capacity_routing_matches_dynamic_and_flags_overflow():
    """The fixed-capacity layout: every occurrence at owner*cap + its rank inside the bucket,
    empty slots id -1; a bucket that does not fit raises the overflow flag."""
    g = torch.Generator().manual_seed(5)
    world, R = 4, 4000
    rows = torch.randint(0, R, (1000,), generator=g)
    st = rd.ShardedTable(R, 4, 0, world, "cpu", cpu_gather, rd.route_torch, capacity_factor=1.2)
    cap = st.capacity(1000)
    assert cap % 64 == 0 and cap >= 300
    pos, counts, send, over = rd.route_torch(rows.view(-1, 1), ZOFF, world, cap)
    assert int(over) == 0 and send.numel() == world * cap
    assert torch.equal(pos // cap, rows % world)              # right bucket
    assert torch.equal(send[pos], rows // world)              # right local row
    assert int((send >= 0).sum()) == len(torch.unique(pos)) == 1000
    dpos, dcounts, dsend = rd.route_torch(rows.view(-1, 1), ZOFF, world)
    starts = torch.cumsum(dcounts, 0) - dcounts
    assert torch.equal(pos % cap, dpos - starts[rows % world])  # same order inside each bucket
    skew = torch.zeros(1000, dtype=torch.int64)                # every occurrence on rank 0
    _, _, _, over = rd.route_torch(skew.view(-1, 1), ZOFF, world, cap)
    assert int(over) == 1


def _worker_padded(rank, world, port, R, D, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.randn(R, D, generator=torch.Generator().manual_seed(123))
        st = rd.ShardedTable(R, D, rank, world, "cpu", cpu_gather, rd.route_torch, capacity_factor=1.3)
        st.load_global(full)
        all_rows = [torch.randint(0, R, (n,), generator=torch.Generator().manual_seed(10 + r))
                    for r in range(world)]
        rows = all_rows[rank]
        buck, ex = st.lookup(rows.view(-1, 1), ZOFF)
        assert ex.cap > 0 and buck.shape[0] == world * ex.cap and int(ex.overflow) == 0
        assert torch.equal(buck[ex.pos][:, :D], full[rows]), "padded lookup mismatch"
        grads = [torch.cat([full[rw] * (r + 1), torch.ones(len(rw), rd.PAD)], 1) for r, rw in enumerate(all_rows)]
        bucketed = torch.full((world * ex.cap, D + rd.PAD), float("nan"))  # empty slots: never read
        bucketed[ex.pos] = grads[rank]
        ids, grows = st.push_grads(ex, bucketed)
        live = ids >= 0
        mine = torch.zeros(st.shard.shape[0], D + rd.PAD, dtype=torch.float64).index_add_(
            0, ids[live], grows[live].double())
        want = torch.zeros(R, D + rd.PAD, dtype=torch.float64)
        for rw, gr in zip(all_rows, grads):
            want.index_add_(0, rw, gr.double())
        assert torch.allclose(mine, want[rank::world], atol=1e-9), "padded gradient push mismatch"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_fixed_capacity_exchange(world):
    port = 29640 + world
    mp.spawn(_worker_padded, args=(world, port, 501, 4, 200), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_lookup_push_allreduce(world):
    port = 29500 + os.getpid() % 2000 + world
    mp.spawn(_worker, args=(world, port, 301, 8, 200), nprocs=world, join=True)
