"""Host logic of the split-graph geometry plan (lion_amd/geometry.py: external / _adopt / _take): the re-keying of an
externally computed FPS / ball-query plan onto a forward's own coordinate tensor and the stage callbacks a chain runner cuts
its graphs at.  Pure bookkeeping -- CPU tensors stand in for the static device buffers."""
import torch

from lion_amd import geometry


def _fake_plan():
    root = torch.zeros(2, 3, 16)
    c1, c2 = torch.zeros(2, 3, 8), torch.zeros(2, 3, 4)
    i1, i2 = torch.zeros(2, 8, 4, dtype=torch.int32), torch.zeros(2, 4, 4, dtype=torch.int32)
    plan = {"fps": {(id(root), 8): (root, c1, 0), (id(c1), 4): (c1, c2, 1)},
            "bq": {(id(c1), id(root), 0.1, 4): (c1, root, i1, 0), (id(c2), id(c1), 0.2, 4): (c2, c1, i2, 1)},
            "root": root, "stages": 2}
    return plan, root, c1, c2, i1, i2


def test_adopted_plan_is_keyed_on_the_forwards_tensor_and_reports_stages_in_order():
    plan, root, c1, c2, i1, i2 = _fake_plan()
    calls = []
    mine = torch.ones(2, 3, 16)                      # the forward's own coordinates: another object, same shape
    adopted = geometry._adopt((plan, lambda first, last: calls.append((first, last))), mine)
    assert adopted is not None and adopted["seen"] == -1
    saved = geometry._PLAN
    geometry._PLAN = adopted
    try:
        assert geometry.lookup_fps(mine, 8) is c1            # stage 0, first use
        assert geometry.lookup_ball_query(c1, mine, 0.1, 4) is i1   # stage 0 again: no second callback
        assert calls == [(0, 0)]
        assert geometry.lookup_fps(c1, 4) is c2              # stage 1
        assert geometry.lookup_ball_query(c2, c1, 0.2, 4) is i2
        assert calls == [(0, 0), (1, 1)]
        assert geometry.lookup_fps(mine, 5) is None          # a miss computes in line
    finally:
        geometry._PLAN = saved


def test_adopt_rejects_other_shapes_and_skipped_stages_are_reported_together():
    plan, root, c1, c2, i1, i2 = _fake_plan()
    assert geometry._adopt((plan, None), torch.ones(2, 3, 17)) is None
    calls = []
    mine = torch.ones(2, 3, 16)
    adopted = geometry._adopt((plan, lambda first, last: calls.append((first, last))), mine)
    saved = geometry._PLAN
    geometry._PLAN = adopted
    try:
        assert geometry.lookup_fps(c1, 4) is c2              # a forward that starts at stage 1: stages 0 .. 1 become due
        assert calls == [(0, 1)]
    finally:
        geometry._PLAN = saved


def test_external_context_nests_and_restores():
    plan, *_ = _fake_plan()
    assert geometry._EXTERNAL is None
    with geometry.external(plan, None):
        assert geometry._EXTERNAL[0] is plan
        with geometry.external({"x": 1}, None):
            assert geometry._EXTERNAL[0] == {"x": 1}
        assert geometry._EXTERNAL[0] is plan
    assert geometry._EXTERNAL is None
