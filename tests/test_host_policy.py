"""Host-side sizing / scheduling policies of the Python layer (no GPU): how check="lazy" calls learn the instance count, the overflow
flag and the long-list chunk count from the pinned header mirror (S360Params.header_mirror — replaces upstream's synchronous
read-back of num_rendered inside every forward), and when S360_FLAG_SPLIT_LISTS is set adaptively."""
import warnings

import torch

from splatter360_amd import rasterizer as R


def _fake_mirror(key, word, cand=0, wide=-1):
    R._MIRRORS[key] = torch.tensor([word, cand, wide], dtype=torch.int64)      # (the real one is pinned; the policy only reads / clears it)


def test_lazy_capacity_follows_the_mirror_with_one_call_of_delay(monkeypatch):
    key = R._hint_key("cpu", 1000, 6, 64, 64, True)
    for d in (R._CAPACITY_HINT, R._CHUNK_HINT, R._MIRRORS, R._OVERFLOW_WARNED, R._SPLIT_AGE):
        d.pop(key, None)
    monkeypatch.setattr(R, "LAZY_SHRINK", True)                          # the opt-in sizing (bench.py's): 1.25 x the largest count seen
    monkeypatch.setattr(R, "DETERMINISTIC", False)
    first = R.default_capacity(1000, 6, 64, 64, device="cpu", lean=True, lazy=True)
    assert first == (3 * 1000 * 6) // 2 + (1 << 18)                      # no count yet: the first-call guess
    _fake_mirror(key, -1)
    assert R.default_capacity(1000, 6, 64, 64, device="cpu", lean=True, lazy=True) == first       # nothing reported yet
    _fake_mirror(key, 5000 | (7 << 33))                                  # 5 000 instances, no overflow, 7 long-list chunks
    cap = R.default_capacity(1000, 6, 64, 64, device="cpu", lean=True, lazy=True)
    assert cap == max(1 << 16, 5000 + 5000 // 4 + (1 << 16))
    assert R.default_segments(key) == R.SEG_PER_CHUNK * (2 * 7 + 64)
    _fake_mirror(key, 3000)                                              # a sparser scene never shrinks the buffers (running maximum)
    assert R.default_capacity(1000, 6, 64, 64, device="cpu", lean=True, lazy=True) == cap
    _fake_mirror(key, 400000 | (1 << 32))                                # a call that overflowed: warned once, next call sized for it
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        big = R.default_capacity(1000, 6, 64, 64, device="cpu", lean=True, lazy=True)
        R.default_capacity(1000, 6, 64, 64, device="cpu", lean=True, lazy=True)
    assert big >= 400000 * 5 // 4 and sum("truncated" in str(x.message) for x in w) == 1
    assert R.overflow_events() >= 1                                      # what a training loop polls to learn of a truncated step
    # check="sync" sizes from the same hint but never polls the mirror itself
    assert R.default_capacity(1000, 6, 64, 64, device="cpu", lean=True, lazy=False) == big


def test_lazy_calls_never_shrink_below_the_first_call_guess_by_default(monkeypatch):
    """a check="lazy" call cannot be re-rendered, so unless the caller opts in (LAZY_SHRINK) it is never sized
    below the first-call guess 1.5 P V; check="sync" calls (re-rendered on overflow) follow the count either way."""
    monkeypatch.setattr(R, "LAZY_SHRINK", False)
    monkeypatch.setattr(R, "DETERMINISTIC", False)
    key = R._hint_key("cpu", 1001, 6, 64, 64, True)
    for d in (R._CAPACITY_HINT, R._CHUNK_HINT, R._MIRRORS, R._OVERFLOW_WARNED, R._SPLIT_AGE):
        d.pop(key, None)
    first = R.default_capacity(1001, 6, 64, 64, device="cpu", lean=True, lazy=True)
    _fake_mirror(key, 5000)
    assert R.default_capacity(1001, 6, 64, 64, device="cpu", lean=True, lazy=True) == first
    assert R.default_capacity(1001, 6, 64, 64, device="cpu", lean=True, lazy=False) == 5000 + 5000 // 4 + (1 << 16)
    _fake_mirror(key, 4 * first)                                         # a denser scene still raises it
    assert R.default_capacity(1001, 6, 64, 64, device="cpu", lean=True, lazy=True) == 4 * first + first + (1 << 16)


def test_deterministic_switch_pins_the_adaptive_choices(monkeypatch):
    monkeypatch.setattr(R, "DETERMINISTIC", True)
    monkeypatch.setattr(R, "LAZY_SHRINK", True)
    key = R._hint_key("cpu", 1002, 6, 64, 64, True)
    for d in (R._CAPACITY_HINT, R._CHUNK_HINT, R._MIRRORS, R._OVERFLOW_WARNED, R._SPLIT_AGE):
        d.pop(key, None)
    assert R.split_decision(key, "auto") is True                         # no report needed: the SPLIT instances, always
    assert R.split_decision(key, False) is False and R.split_decision(key, "auto", quadrant_waves=R.AUTO_SPLIT_MAX_WAVES + 4) is False
    first = R.default_capacity(1002, 6, 64, 64, device="cpu", lean=True, lazy=True)
    _fake_mirror(key, 5000 | (7 << 33))
    assert R.default_capacity(1002, 6, 64, 64, device="cpu", lean=True, lazy=True) == first       # never below the first-call guess
    assert R.default_segments(key) == 0                                  # the library's worst-case segment storage, not a history-sized one


def test_split_flag_is_adaptive_by_default_and_can_be_forced(monkeypatch):
    monkeypatch.setattr(R, "DETERMINISTIC", False)
    key = R._hint_key("cpu", 2000, 6, 64, 64, True)
    R._MIRRORS.pop(key, None); R._SPLIT_AGE.pop(key, None)
    assert R.split_decision(key, True) is True and R.split_decision(key, False) is False
    assert R.split_decision(key, "auto") is False                        # nothing reported: the kernels without the hand-over code
    _fake_mirror(key, 100, cand=1)                                       # a forward reported a quadrant worth splitting
    assert R.split_decision(key, "auto") is True and int(R._MIRRORS[key][1]) == 0      # ... consumed
    for _ in range(16):
        assert R.split_decision(key, "auto") is True                     # stays on for 16 calls without a new report
    assert R.split_decision(key, "auto") is False
    R._MIRRORS[key][1] = 1
    assert R.split_decision(key, "auto") is True
    # calls whose (tile, quadrant) waves fill the chip more than four times never split adaptively (second launch = after ALL tiles)
    R._MIRRORS[key][1] = 1
    big = R.AUTO_SPLIT_MAX_WAVES + 4     # beyond the largest shape splitting was measured to pay at (4 M Gaussians, six 512^2 faces)
    assert R.split_decision(key, "auto", quadrant_waves=big) is False and R.split_decision(key, True, quadrant_waves=big) is True


def test_coop_walk_follows_the_wide_rectangle_count_of_the_previous_call():
    key = R._hint_key(None, 1004, 6, 64, 64, True)
    R._MIRRORS.pop(key, None)
    assert R.coop_decision(key, "auto") is False                          # nothing known yet: the default kernels
    _fake_mirror(key, 100, wide=300)                                      # the headline cloud's order of magnitude
    assert R.coop_decision(key, "auto") is False
    _fake_mirror(key, 100, wide=R.AUTO_COOP_MIN_PAIRS)                    # a cloud of near, screen-filling splats
    assert R.coop_decision(key, "auto") is True
    assert R.coop_decision(key, False) is False and R.coop_decision(key, True) is True
    R._MIRRORS.pop(key, None)
