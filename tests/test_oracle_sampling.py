"""The oracle's logits processors (oracle/sampling.py = framework/sampling/logits_utils.cpp restated) on hand-computed cases: the
reference holds no vector for them, so the semantics the HIP kernels are held to are pinned here by inspection-sized examples."""
import math

import torch

from oracle import sampling as osm

NINF = float("-inf")


def test_penalties_by_hand():
    logits = torch.tensor([[1.0, -2.0, 3.0, 0.5], [0.0, 4.0, -1.0, 2.0]])
    ids = torch.tensor([[2, 1, 0, 0], [3, 0, 0, 0]])            # padded with id 0 / count 0 (sampling_params.cpp:127-134)
    cnt = torch.tensor([[2, 1, 0, 0], [3, 0, 0, 0]], dtype=torch.int32)
    osm.apply_frequency_presence_penalties(logits, ids, cnt, torch.tensor([0.5, 1.0]), torch.tensor([0.25, 0.0]))
    # row 0: col 2: 3 - 2*0.5 - 0.25 = 1.75; col 1: -2 - 0.5 - 0.25 = -2.75; col 0 (count 0): unchanged
    assert logits.tolist() == [[1.0, -2.75, 1.75, 0.5], [0.0, 4.0, -1.0, -1.0]]
    osm.apply_repetition_penalties(logits, ids, torch.tensor([2.0, 4.0]))
    # every listed id, padding included: positive / 2, negative * 2 -- col 0 of row 0 (padding id) is divided ONCE
    assert logits.tolist() == [[0.5, -5.5, 0.875, 0.5], [0.0, 4.0, -1.0, -4.0]]


def test_temperature_zero_means_one():
    logits = torch.tensor([[2.0, 4.0], [2.0, 4.0]])
    osm.apply_temperatures(logits, torch.tensor([0.0, 2.0]))
    assert logits.tolist() == [[2.0, 4.0], [1.0, 2.0]]


def test_top_k_and_top_p_rules_by_hand():
    lg = lambda p: [math.log(x) for x in p]
    row = torch.tensor([lg([0.1, 0.4, 0.2, 0.3])])              # sorted ranks: col 1 (.4), col 3 (.3), col 2 (.2), col 0 (.1)
    # one of them -- top-p, EXCLUSIVE prefix: ranks with (cum - prob) > p are masked: prefixes 0, .4, .7, .9
    out = osm.apply_top_k_top_p(row.clone(), None, None, torch.tensor([0.5]))
    assert [v == NINF for v in out[0].tolist()] == [True, False, True, False]          # ranks 0, 1 (cols 1, 3) survive (.4 <= .5 < .7)
    out = osm.apply_top_k_top_p(row.clone(), None, None, torch.tensor([0.0]))
    assert [v == NINF for v in out[0].tolist()] == [True, False, True, True]           # rank 0 always survives
    # one of them -- top-k, k <= 0 disables
    out = osm.apply_top_k_top_p(row.clone(), None, torch.tensor([2]), None)
    assert [v == NINF for v in out[0].tolist()] == [True, False, True, False]
    assert torch.equal(osm.apply_top_k_top_p(row.clone(), None, torch.tensor([0]), None), row)
    # both -- clamp(k, 1, V), INCLUSIVE prefix, rank 0 forced: top-3 renormalised = .4/.9, .3/.9, .2/.9 -> cum .444, .778, 1.0
    both = row.clone()
    osm.apply_top_k_top_p(both, None, torch.tensor([3]), torch.tensor([0.8]))
    assert [v == NINF for v in both[0].tolist()] == [True, False, True, False]         # cum <= .8: ranks 0, 1
    # both, k <= 0 (the reference's default top_k = -1 in a batch where ANOTHER request set top_k): no limit from k -- the row is
    # filtered by p alone (inclusive prefixes .4, .7, .9, 1.0; rank 0 forced). Round-4 advisor: a clamp to 1 made it greedy.
    both = row.clone()
    osm.apply_top_k_top_p(both, None, torch.tensor([0]), torch.tensor([0.1]))
    assert [v == NINF for v in both[0].tolist()] == [True, False, True, True]          # p = .1: only the forced rank 0
    both = row.clone()
    osm.apply_top_k_top_p(both, None, torch.tensor([-1]), torch.tensor([0.75]))
    assert [v == NINF for v in both[0].tolist()] == [True, False, True, False]         # cum <= .75: ranks 0, 1 -- NOT top-1
    both = torch.cat([row, row])                                                       # a mixed batch: k = 2 and k = -1 side by side
    osm.apply_top_k_top_p(both, None, torch.tensor([2, -1]), torch.tensor([0.95, 0.95]))
    assert [v == NINF for v in both[0].tolist()] == [True, False, True, True]          # top-2 renormalised: cum .571, 1.0 -> only rank 0 is <= .95
    assert [v == NINF for v in both[1].tolist()] == [True, False, False, False]        # no k: cum .4, .7, .9 <= .95 -> ranks 0, 1, 2
    # the reference's torch_impl as written still clamps (restated faithfully; it is not what runs on CUDA / DCU / NPU)
    both = row.clone()
    osm.apply_top_k_top_p_torch_impl(both, torch.tensor([0]), torch.tensor([0.99]))
    assert [v == NINF for v in both[0].tolist()] == [True, False, True, True]
    # ties rank by column index (stable sort)
    tie = torch.tensor([[1.0, 2.0, 2.0, 2.0]])
    out = osm.apply_top_k_top_p(tie.clone(), None, torch.tensor([2]), None)
    assert [v == NINF for v in out[0].tolist()] == [True, False, False, True]
