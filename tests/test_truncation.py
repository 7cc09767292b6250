import warnings

import numpy as np

from helpers import golden
from tenpy_amd.linalg.truncation import truncate, TruncationError


def test_truncate_golden():
    for rec in golden('truncate.pkl'):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            mask, norm_new, err = truncate(rec['S'], dict(rec['options']))
        np.testing.assert_array_equal(mask, rec['mask'])
        assert norm_new == rec['norm_new'] and err.eps == rec['eps']


def test_truncation_error():
    e = TruncationError.from_S(np.array([0.1, 0.2])) + TruncationError(0.01, 0.98)
    assert abs(e.eps - 0.06) < 1e-15 and abs(e.ov - (1 - 0.1) * 0.98) < 1e-15
