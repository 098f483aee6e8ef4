"""Row a1: the embedding constants are data and must be reproduced digit for digit."""
import json
import os

import numpy as np

from hsearch_amd import synth


def test_coords_match_reference_dump(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "constants.json")))
    assert np.array_equal(synth.coords(), np.array(g["coordinates"]))


def test_distance_square_known_answer(golden_dir):
    # util.hpp:43-64 is the 6-decimal print of |coord_i - coord_j|^2 (max diff 4.93e-7)
    g = json.load(open(os.path.join(golden_dir, "constants.json")))
    c = synth.coords()
    d2 = ((c[:, None, :] - c[None, :, :]) ** 2).sum(-1)
    assert np.abs(d2 - np.array(g["DISTANCE_SQUARE"])).max() < 5e-7


def test_letter_map_matches_base(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "constants.json")))
    letters = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
    codes, unknown = oracle.letters_to_codes(letters)
    base = np.array(g["base"])
    assert unknown == int((base < 0).sum()) == 6
    for i, ch in enumerate(letters):
        assert (codes[i] == 255) == (base[i] < 0)
        if base[i] >= 0:
            assert codes[i] == base[i]
    # BLOSUM order: Q is row 5, E is row 6 (SURVEY appendix, E/Q swap of AA20)
    from hsearch_amd import alphabet, codes_from_letters
    assert alphabet()[5] == "Q" and alphabet()[6] == "E"
    assert np.array_equal(codes_from_letters(["AQE"])[0], [0, 5, 6])
