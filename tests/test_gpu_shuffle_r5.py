"""Round 5: hypad_epoch_shuffles sorts its Philox keys with register-resident compare-exchange stages (blocks of 128 keys per wave) and
only the wide stages through LDS.  Element i's key depends on (seed, tick, pass, i) alone, so the permutations of different window counts
are ONE order of the keys: the permutation of n windows is the permutation of n' > n windows with the entries >= n struck out.  Held here
across every code path: fewer than 128 keys (the LDS-only sort of rounds 3-4), one block per wave (2 048 keys), two (4 096)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_permutations_of_all_window_counts_are_one_order_of_the_keys():
    from hypad_amd import _C
    counters = torch.zeros(8, dtype=torch.int32, device="cuda")
    counts = (40, 64, 65, 127, 128, 129, 700, 1916, 2047, 2048, 2049, 3000, 4095, 4096)
    for tick in (0, 7):
        counters[3] = tick
        perms = {}
        for n in counts:
            out = torch.full((3, n), -1, dtype=torch.int32, device="cuda")
            _C.check(_C.lib.hypad_epoch_shuffles(out.data_ptr(), 3, n, n, 99, counters.data_ptr(), _C.stream()), "epoch_shuffles")
            torch.cuda.synchronize()
            perms[n] = out.cpu()
            for p in range(3):
                assert sorted(perms[n][p].tolist()) == list(range(n)), (n, p)
        big = perms[4096]
        for n in counts[:-1]:
            for p in range(3):
                sub = big[p][big[p] < n]
                assert torch.equal(sub, perms[n][p]), (tick, n, p)
        assert not torch.equal(perms[4096][0], perms[4096][1])
