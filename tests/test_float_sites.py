"""The reference computes three things in f32 (bc7.rs:408-553, etc.rs:297-307).  The GPU path uses
integer forms; these tests prove the two agree on every reachable input,
using the oracle's f32-faithful restatement as the truth."""


def test_shared_pbits_integer_equals_f32_on_the_whole_uastc_mode2_domain(oracle):
    # exhaustive 16^6 endpoint pairs (UASTC mode 2 endpoints are multiples of 17)
    assert oracle.lib.bu_oracle_prove_shared_pbits(None) == 0


def test_unique_pbits_integer_equals_f32(oracle):
    # BC7 mode 6 (4 comps, 7+1 bits), mode 3 (3 comps, 7+1 bits), mode 7 (4 comps, 5+1 bits)
    for comps, bits in ((4, 7), (3, 7), (4, 5)):
        assert oracle.lib.bu_oracle_prove_unique_pbits(comps, bits, 3_000_000, 1234 + comps + bits) == 0


def test_eac_centre_integer_equals_f32(oracle):
    # all 16 modifier tables x all 0 <= min < max <= 255
    assert oracle.lib.bu_oracle_prove_eac_center() == 0
