"""CPU, world_size 2, gloo: the multi-GPU path is a contiguous batch split with one
all_gather of the results; x_T is drawn for the global batch, so the gathered output is
identical to the single-rank output for any number of ranks."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def test_shard_range_partitions():
    sr = pkg("shard").shard_range
    for total in (0, 1, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            spans = [sr(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sr(4, 2, 2)


def _fake_enhance(wav, x_T):
    # per-utterance function of (wav, x_T): stands in for the GPU path on CPU ranks
    return wav * 2.0 + x_T.flatten(1)[:, : wav.shape[1]].mean(dim=1, keepdim=True)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import pkg as _pkg

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wav, x_T = _pkg("synth").synthetic_waveforms(5, 320, seed=3)      # ragged: 5 utterances over 2 ranks
    out = _pkg("shard").enhance_sharded(_fake_enhance, wav, x_T)
    torch.save(out, os.path.join(tmp, "out%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_rank(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    wav, x_T = pkg("synth").synthetic_waveforms(5, 320, seed=3)
    ref = _fake_enhance(wav, x_T)
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), "out%d.pt" % r))
        assert torch.equal(got, ref)
    assert torch.equal(pkg("shard").enhance_sharded(_fake_enhance, wav, x_T), ref)
