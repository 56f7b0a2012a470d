"""Arithmetic of bench.py's roofline block: the contract figure, the unique-bytes figure and the
counter-traffic figure over one measured kernel time, and the choice of `bound`."""
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
    assert abs(rf["achieved"] - b_eval * B / 50e-6 / 1e9) < 0.01
    assert abs(rf["frac"] - b_eval * B / 50e-6 / 8e12) < 1e-5 and rf["frac"] > 1.0            # the contract figure exceeds 1
    assert abs(rf["frac_unique"] - unique / 50e-6 / 8e12) < 1e-5 and rf["frac_unique"] < 0.5
    assert abs(rf["frac_traffic"] - 150e6 / 50e-6 / 8e12) < 1e-5
    assert rf["unique_bytes_per_launch"] == unique and rf["traffic"] == 150e6
    assert rf["bound"] == "valu"                                      # 0.55 of vector issue > 0.375 of HBM
    mp = rf["matrix_pipe"]
    assert mp["int8_ops_per_launch"] == 2.0 * (128 * 96) * 2016 * 1024
    assert abs(mp["frac"] - mp["int8_ops_per_launch"] / 50e-6 / 5e15) < 1e-4
    # no counter pass of this build: frac_traffic withheld; the vector figure still decides against the unique-bytes figure
    rf2 = bench.roofline_block(b_eval, unique, B, kern_ms, None, valu, "k_mixture_tuple64<...>", True, "test")
    assert rf2["frac_traffic"] is None and rf2["traffic"] is None and rf2["bound"] == "valu" and "matrix_pipe" not in rf2
    # a streaming kernel: traffic fraction above the vector fraction -> hbm
    rf3 = bench.roofline_block(4629008, 64 * 4629008, 64, 0.1466, {"bytes_per_launch": 248.6e6, "source": "t"}, {"frac": 0.15}, "k_mixture_rows<...>", True, "t")
    assert rf3["bound"] == "hbm" and abs(rf3["frac_traffic"] - 248.6e6 / 146.6e-6 / 8e12) < 1e-5
    # nothing static at all
    assert bench.roofline_block(b_eval, unique, B, kern_ms, None, None, "k", True, "t")["bound"] == "hbm"
