"""CPU: the decode oracle is pinned against goldens produced by the imported reference."""
import numpy as np
import pytest

from decode_cases import cases
from oracle import decode as oracle_decode

GOLD = np.load(__file__.replace("test_decode_oracle.py", "golden/decode_goldens.npz"))
CASES = list(cases())


@pytest.mark.parametrize("name,hm,box,thr", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_goldens(name, hm, box, thr):
    got = oracle_decode.get_keypoints(hm, box, thr)
    want = GOLD[f"{name}/keypoints"]
    assert got.dtype == np.int32 and got.shape == (17, 3)
    np.testing.assert_array_equal(got, want)


def test_all_golden_names_covered():
    assert sorted(GOLD["names"].tolist()) == sorted(c[0] for c in CASES)


def test_survey_known_answers():
    # values observed from the reference during the survey
    np.testing.assert_array_equal(GOLD["rand128_fullbox/keypoints"][:5],
                                  [[44, 208, 1], [24, 380, 1], [96, 140, 1], [384, 12, 1], [308, 204, 1]])
    np.testing.assert_array_equal(GOLD["trunc_8x6/keypoints"][0], [20, 28, 1])
    np.testing.assert_array_equal(GOLD["offset_ignored/keypoints"], GOLD["trunc_8x6/keypoints"])
    np.testing.assert_array_equal(GOLD["inverted_box/keypoints"][0], [-5, -5, 1])
    np.testing.assert_array_equal(GOLD["plus_inf/keypoints"][0], [37, 50, 1])


def test_scores_and_indices_tie_rule():
    x = np.zeros((2, 3, 4, 17), np.float32)
    x[0, 1, 2, :] = 1.0
    x[0, 2, 3, :] = 1.0
    mx, idx = oracle_decode.scores_and_indices(x)
    assert (idx[0] == 1 * 4 + 2).all() and (idx[1] == 0).all()
    assert (mx[0] == 1).all()
