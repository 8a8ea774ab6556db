"""train.plan_signal_groups (CPU): which rank trains which signal, in which launch groups, with which random stream."""
import pytest

from hypad_amd.train import plan_signal_groups


@pytest.mark.parametrize("counts,batch", [([1916, 700, 1950, 1916, 640, 1920, 3000], 64), ([1916] * 64, 64), ([300, 260, 520, 513, 256], 256),
                                          ([64], 64), (list(range(64, 64 + 37 * 7, 7)), 64)])
@pytest.mark.parametrize("group", [1, 3, 32])
def test_every_signal_once_streams_do_not_depend_on_the_world(counts, batch, group):
    ref_stream = None
    for world in (1, 2, 3, 8):
        seen, loads = [], []
        for rank in range(world):
            groups, stream = plan_signal_groups(counts, batch, world, rank, group)
            if ref_stream is None:
                ref_stream = stream
            assert stream == ref_stream                                   # a signal's stream number: a property of the call, not of the sharding
            mine = 0
            for first, members in groups:
                assert 1 <= len(members) <= group
                assert len({counts[i] // batch for i in members}) == 1     # one batch count per launch group
                assert [stream[i] for i in members] == list(range(first, first + len(members)))      # contiguous streams: Engine.first_signal + slot
                seen += members
                mine += len(members)
            loads.append(mine)
        assert sorted(seen) == list(range(len(counts)))
        assert max(loads) - min(loads) <= len({c // batch for c in counts})      # balanced up to one signal per batch-count run
    # streams are the ranks in (batch count, position) order
    order = sorted(range(len(counts)), key=lambda i: (counts[i] // batch, i))
    assert [ref_stream[i] for i in order] == list(range(len(counts)))


def test_a_signal_too_short_for_one_batch_is_refused():
    from hypad_amd._C import HypadError
    with pytest.raises(HypadError, match="do not fill one batch"):
        plan_signal_groups([640, 63], 64)
