"""Randomised small shapes: GPU vs oracle for odd N / M / K, heavy missingness and flips
(ragged tiles, single variants, samples < one wave, all-missing and monomorphic rows)."""
import numpy as np
import pytest

from conftest import assert_table_close

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method="thread")]


def _case(rng, n, m, k, trait, miss):
    from saigegds_amd import synth
    from saigegds_amd.gds import pack_dosage_2bit
    from saigegds_amd.nullmod import init_nullmod
    mod = synth.synth_null_model(n, trait, 0.3, n_cov=k, seed=int(rng.integers(1, 1 << 30)))
    sm = init_nullmod(mod, np.arange(n), float("nan"), 1, 0.5, 0.05, float(mod.var_ratio[0]))
    p = rng.uniform(0.02, 0.98, size=m)
    codes = (rng.random((m, n)) < p[:, None]).astype(np.uint8) + (rng.random((m, n)) < p[:, None]).astype(np.uint8)
    codes[rng.random((m, n)) < miss] = 3
    if m > 2:
        codes[0] = 3            # all missing
        codes[1] = 0            # monomorphic
        codes[2] = 2
    return sm, pack_dosage_2bit(codes)


@pytest.mark.parametrize("seed", range(8))
def test_random_small_shapes(seed):
    import torch  # noqa: F401
    from oracle import Oracle
    from saigegds_amd._lib import Scanner
    rng = np.random.default_rng(1000 + seed)
    for _ in range(5):
        n = int(rng.choice([40, 63, 64, 65, 200, 255, 257, 511, 513, 700]))   # (a handful of samples with K covariates leaves var2 = 0 up to rounding: nothing to compare)
        m = int(rng.choice([1, 2, 3, 15, 16, 17, 100, 257]))
        k = int(rng.integers(1, 7))
        trait = "binary" if rng.random() < 0.7 else "quantitative"
        miss = float(rng.choice([0.0, 0.01, 0.3]))
        if trait == "binary" and n < 60:
            continue            # the synthetic logistic fit needs a few cases and controls
        sm, packed = _case(rng, n, m, k, trait, miss)
        ref, ref_valid = Oracle(sm).scan_2bit(packed)
        with Scanner(sm) as sc:
            out, valid = sc.scan_2bit(packed)
            sc.set_option("lanes", 2)
            out2, valid2 = sc.scan_2bit(packed)
        what = f"seed {seed}: N={n} M={m} K={k} {trait} miss={miss}"
        assert_table_close(out, valid, ref, ref_valid, quant=sm.quant, what=what)
        assert np.array_equal(valid, valid2) and np.array_equal(np.nan_to_num(out, nan=-7), np.nan_to_num(out2, nan=-7)), what


def test_twelve_fragments_block_and_rows_give_the_same_table():
    """K = 12 fills 11 value fragments (+ bit-1 = 12 B fragments of the contraction kernel): the row-major call and
    a resident block of the same rows give the same table bit for bit (integer accumulation; the SPA stage reads
    the same rows either way, the block's carrier lists hold the same carriers), both within the usual bounds of
    the oracle."""
    import torch
    from oracle import Oracle
    from saigegds_amd._lib import Block, Scanner
    rng = np.random.default_rng(77)
    sm, packed = _case(rng, 3000, 700, 12, "binary", 0.01)
    ref, ref_valid = Oracle(sm).scan_2bit(packed)
    dev = torch.device("cuda", 0)
    with Scanner(sm) as sc, Block(sm.n, 700) as blk:
        limbs, ngroups = sc.score_layout()
        ncol = int(sum(limbs)) + 1
        assert 160 < ncol <= 176, f"the case is meant to fill 11 value fragments (+ bit-1 = 12): {ncol} limb columns"
        out, valid = sc.scan_2bit(packed)
        sc.load_block(blk, packed)
        o2 = torch.full((700, 8), -1.0, dtype=torch.float64, device=dev)
        v2 = torch.zeros(700, dtype=torch.uint8, device=dev)
        sc.set_option("spa_abl", 512)            # the SPA kernels scan the rows, as the row-major call does
        sc.scan_block(blk, o2.data_ptr(), v2.data_ptr())
        sc.sync()
    assert np.array_equal(valid, v2.cpu().numpy())
    assert np.array_equal(np.nan_to_num(out, nan=-7), np.nan_to_num(o2.cpu().numpy(), nan=-7))
    assert_table_close(out, valid, ref, ref_valid, quant=False, what="K = 12")
