"""nanomod_amd — MI355X (gfx950) implementation of NanoMod's per-base two-sample
testing hot path (KS / Mann-Whitney U / Welch t per genomic position + sliding
window Stouffer / Fisher combine), behind the reference's own Python call
boundary.  See DESIGN.md and INTEGRATION.md."""
from . import _lib  # noqa: F401
from .detect import (mtest2, mfilter_coverage, getKStest, get_combin_pvalue, combin_pvalues,  # noqa: F401
                     pos_check, save_test, m_min_float, m_max_float, run_ids, build_csr, encode_signals, region_rank)
from .engine import detect_host, combine_host, DeviceDetector, EventTimer  # noqa: F401
from . import simulate, sharding  # noqa: F401

__all__ = ['mtest2', 'mfilter_coverage', 'getKStest', 'get_combin_pvalue', 'combin_pvalues', 'pos_check',
           'save_test', 'm_min_float', 'm_max_float', 'run_ids', 'build_csr', 'encode_signals',
           'detect_host', 'combine_host', 'DeviceDetector', 'EventTimer']
