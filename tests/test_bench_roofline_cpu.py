"""Arithmetic of bench.py's roofline block: the contract figure, the unique-bytes figure and the
counter-traffic figure over one measured kernel time, and the choice of `limiter` (`bound` is the roofline the figures are priced against: hbm)."""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

import bench                                                          # noqa: E402
from sbayes_amd.synthetic import algorithmic_bytes, unique_bytes_per_launch      # noqa: E402


def test_unique_bytes_count_the_shared_block_once():
    n, f, s, groups, p, B = 1000, 200, 10, [5, 1], 2, 2048
    b_eval = algorithmic_bytes(n, f, s, groups, p, packed=True)
    assert b_eval == 1000 * 200 + 6 * 200 * 10 * 4 + 2 * 200 * 2 * 4 + 1000 * 3 + 8 == 254208        # SURVEY.md 8(d), packed
    unique = unique_bytes_per_launch(n, f, s, groups, p, B, packed=True)
    assert unique == 200000 + B * (b_eval - 200000)
    assert unique_bytes_per_launch(n, f, s, groups, p, 1, packed=True) == b_eval
    assert unique_bytes_per_launch(n, f, s, groups, p, B, packed=False) == 2000000 + B * (b_eval - 200000)


def test_roofline_block_fractions_and_bound():
    b_eval, B, kern_ms = 254208, 2048, 0.050
    unique = 200000 + B * (b_eval - 200000)
    traffic = {"bytes_per_launch": 150e6, "source": "test"}
    valu = {"frac": 0.55}
    rf = bench.roofline_block(b_eval, unique, B, kern_ms, traffic, valu, "k_mixture_tuple_mfma<..., M tiles 3, C=2>", True, "test",
                              shape=(1000, 200, 10))
    # `frac` is the unique-bytes figure -- a fraction of the peak, <= 1 -- and `achieved` the same in GB/s
    assert abs(rf["achieved"] - unique / 50e-6 / 1e9) < 0.01
    assert abs(rf["frac"] - unique / 50e-6 / 8e12) < 1e-5 and rf["frac"] < 0.5
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    # the contract figure (shared block counted once per eval) exceeds 1 and lives under its own key
    assert abs(rf["frac_contract"] - b_eval * B / 50e-6 / 8e12) < 1e-5 and rf["frac_contract"] > 1.0
    assert abs(rf["achieved_contract"] - b_eval * B / 50e-6 / 1e9) < 0.01
    assert "frac_unique" not in rf
    assert abs(rf["frac_traffic"] - 150e6 / 50e-6 / 8e12) < 1e-5
    assert rf["unique_bytes_per_launch"] == unique and rf["traffic"] == 150e6
    assert rf["bound"] == "hbm" and rf["limiter"] == "valu"           # priced against HBM; 0.55 of vector issue > 0.375 of HBM limits it
    mp = rf["matrix_pipe"]
    assert mp["int8_ops_per_launch"] == 2.0 * (128 * 96) * 2016 * 1024
    assert abs(mp["frac"] - mp["int8_ops_per_launch"] / 50e-6 / 5e15) < 1e-4
    # FP4 operands, the round-6 name: slots per block parsed from the name, k-blocks of 64 objects, the FP4 peak
    rf4 = bench.roofline_block(b_eval, unique, B, kern_ms, traffic, valu, "k_mixture_tuple_mfma<packed stream, group-tuple form, matrix pipe fp4, 16 slots x M tiles 3, C=2>",
                               True, "test", shape=(1000, 200, 10))
    assert rf4["matrix_pipe"]["fp4_ops_per_launch"] == 2.0 * (128 * 96) * 2016 * 1024 and rf4["matrix_pipe"]["peak_tops"] == 10000.0
    rf5 = bench.roofline_block(b_eval, unique, B, kern_ms, traffic, valu, "k_mixture_tuple_mfma<packed stream, group-tuple form, matrix pipe fp4, 4 slots x M tiles 3, C=3>",
                               True, "test", shape=(100, 36, 5))
    assert rf5["matrix_pipe"]["fp4_ops_per_launch"] == 2.0 * (512 * 96) * 192 * 256
    # no counter pass of this build: frac_traffic withheld; the vector figure still decides against the unique-bytes figure
    rf2 = bench.roofline_block(b_eval, unique, B, kern_ms, None, valu, "k_mixture_tuple64<...>", True, "test")
    assert rf2["frac_traffic"] is None and rf2["traffic"] is None and rf2["limiter"] == "valu" and rf2["bound"] == "hbm" and "matrix_pipe" not in rf2
    # a streaming kernel: traffic fraction above the vector fraction -> hbm
    rf3 = bench.roofline_block(4629008, 64 * 4629008, 64, 0.1466, {"bytes_per_launch": 248.6e6, "source": "t"}, {"frac": 0.15}, "k_mixture_rows<...>", True, "t")
    assert rf3["bound"] == "hbm" and rf3["limiter"] == "hbm" and abs(rf3["frac_traffic"] - 248.6e6 / 146.6e-6 / 8e12) < 1e-5
    assert rf3["frac"] == rf3["frac_contract"]                        # one block per state: nothing is shared, the two agree
    # nothing static at all
    assert bench.roofline_block(b_eval, unique, B, kern_ms, None, None, "k", True, "t")["limiter"] == "hbm"


def test_frac_never_exceeds_one_for_any_kernel_time_the_hardware_allows():
    """A launch cannot move its unique bytes faster than the peak: at the fastest physically possible kernel time
    (unique / 8 TB/s) frac is exactly 1 while the contract figure is far above it."""
    b_eval, B = 254208, 4096
    unique = 200000 + B * (b_eval - 200000)
    t_min_ms = unique / 8e12 * 1e3
    rf = bench.roofline_block(b_eval, unique, B, t_min_ms, None, None, "k", True, "t")
    assert abs(rf["frac"] - 1.0) < 1e-4 and rf["frac_contract"] > 4.0


def test_median_repetition_pairs_time_and_span():
    """value, ms_per_step and the kernel span come from ONE repetition: the median one (upper median for an even count)."""
    assert bench._median_rep([3.0, 1.0, 2.0]) == 2
    assert bench._median_rep([4.0, 1.0, 3.0, 2.0]) == 2
    assert bench._median_rep([5.0]) == 0
