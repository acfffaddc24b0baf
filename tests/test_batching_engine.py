"""BatchEngine protocol (lock-step + one-deep pipelining, slot recycling, prefix cache) against a fake
model: randomised joins / leaves from more threads than slots.  CPU only."""
import random
import threading
import time
from types import SimpleNamespace

import pytest
import torch

from detikzify_amd.infer.batching import BatchEngine


class _FakeLib:
    def __init__(self, m):
        self.m = m

    def dtk_context_len_slot(self, ctx, s):
        return self.m.pos[s]


class _FakeModel:
    """slot s emits tokens s*1000 + 0, 1, 2, ...; asserts the C-ABI call discipline"""

    def __init__(self, n, max_positions=10 ** 9):
        self.n, self.pos, self.cnt, self.launched = n, [0] * 17, [0] * 17, []
        self.config = SimpleNamespace(max_positions=max_positions, image_token_id=1)
        self.lib, self._ctx, self.batch_engine, self.steps, self.prefills, self.forks = _FakeLib(self), None, None, 0, 0, 0

    def num_slots(self):
        return self.n

    def set_sampling(self, slot=None, **kw):
        pass

    def prefill(self, ids, px, slot=None, reuse=None):
        assert not self.launched, "prefill while a step is un-collected (the C side would drop it)"
        self.pos[slot], self.cnt[slot] = len(ids), 0
        self.prefills += 1

    def image_key(self, px):
        return int(px.sum())

    def kv_fork(self, a, b, n):
        self.pos[b], self.cnt[b] = n, 0
        self.forks += 1

    def best_lcp_slot(self, slots, ids, key=0):
        return None        # this device remembers no token ids: nothing to resume in place

    def decode_batch_launch(self, slots):
        assert len(self.launched) < 2 and slots == sorted(slots) and all(0 <= s < 64 for s in slots)
        assert all(self.pos[s] < self.config.max_positions for s in slots)
        self.launched.append(list(slots))
        for s in slots:
            self.pos[s] += 1

    def decode_batch_wait(self):
        time.sleep(0.0003)
        sl = self.launched.pop(0)
        self.steps += 1
        out = [-1] * 64
        for s in sl:
            out[s] = s * 1000 + self.cnt[s]
            self.cnt[s] += 1
        return out


@pytest.mark.parametrize("mode", ["pull", "push", "mixed"])
@pytest.mark.parametrize("pipeline", [True, False])
def test_engine_random_joins_and_leaves(pipeline, mode):
    """pull = seq.next_token() per token in the sequence's own thread; push = seq.run(emit): emit is called by whichever
    thread drives the steps (what model.generate uses); mixed = both kinds in one batch"""
    m = _FakeModel(5)
    eng = BatchEngine(m, max_batch=4, pipeline=pipeline)
    assert eng.share_prefix and eng.prefix_slot == 4 and eng.capacity == 4
    errs, total = [], [0]

    def worker(i):
        rng = random.Random(i)
        try:
            for _ in range(5):
                n = rng.randint(1, 20)
                with eng.sequence(torch.tensor([1, 1, 1, 7][: 3 + (i % 2)]), torch.ones(1), {}) as seq:
                    if mode == "pull" or (mode == "mixed" and i % 2):
                        got = [seq.next_token() for _ in range(n)]
                    else:
                        got = []
                        seq.run(lambda tok: got.append(tok) or len(got) >= n)
                    assert [g % 1000 for g in got] == list(range(n)) and len({g // 1000 for g in got}) == 1
                total[0] += n
                time.sleep(rng.random() * 0.001)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(7)]
    [t.start() for t in ths]
    [t.join(timeout=60) for t in ths]
    assert not any(t.is_alive() for t in ths) and not errs, errs[:1]
    eng.close()
    assert sorted(eng.free) == [0, 1, 2, 3] and not eng.zombies and eng.inflight is None and not m.launched
    assert total[0] <= eng.tokens_out <= total[0] + 35      # a sequence may leave one delivered token unread
    # one prefix prefill (same image); every join either forks the prefix or finds it in its slot already
    assert m.forks + eng.inplace_reuses == 35 and eng.prefix_encodes == 1 and m.prefills <= 36 + eng.inplace_reuses


def test_engine_respects_context_limit_and_errors():
    m = _FakeModel(3, max_positions=6)
    eng = BatchEngine(m, max_batch=2)
    with eng.sequence(torch.tensor([1, 1, 1]), torch.ones(1), {}) as seq:
        assert [seq.next_token() % 1000 for _ in range(3)] == [0, 1, 2]     # positions 3,4,5 -> full
    eng.close()

    class Boom(_FakeModel):
        def decode_batch_wait(self):
            raise RuntimeError("device lost")
    eng = BatchEngine(Boom(3), max_batch=2)
    with pytest.raises(RuntimeError):
        with eng.sequence(torch.tensor([1, 1]), torch.ones(1), {}) as seq:
            seq.next_token()
    with pytest.raises(ValueError):
        BatchEngine(_FakeModel(0))


def test_engine_shares_prefixes_of_several_images_in_flight():
    """8 images x 3 rollouts through a 9-slot engine (one prefix-cache slot): every image is encoded once; later rollouts
    of an image fork the prefix KV from any slot that still holds it, not from a re-encoded prefix cache"""
    m = _FakeModel(9)
    eng = BatchEngine(m, max_batch=8)
    errs = []

    def worker(i):
        try:
            px = torch.full((1,), float(i % 8))                 # image id
            with eng.sequence(torch.tensor([1, 1, 1, 7, 9]), px, {}) as seq:
                got = [seq.next_token() for _ in range(6 + i % 3)]
                assert [g % 1000 for g in got] == list(range(len(got)))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    for wave in range(3):                                       # 3 rollouts per image, 8 images at a time
        ths = [threading.Thread(target=worker, args=(wave * 8 + j,)) for j in range(8)]
        [t.start() for t in ths]
        [t.join(timeout=60) for t in ths]
        assert not any(t.is_alive() for t in ths) and not errs, errs[:1]
    eng.close()
    assert eng.prefix_encodes == 8, eng.prefix_encodes          # not 24: waves 2 and 3 find a donor slot per image
    assert m.forks + eng.inplace_reuses == 24
