"""msom_amd -- MI355X-native multi-layer quasi-geostrophic timestepper.

The product is the HIP library ``msom_amd/lib/libmsomhip.so`` (C ABI: ``include/msom.h``).
This package is the thin Python host mirror of the reference's SWIG module ``qg``
(msqg/qg.i, msqg/qg_bfn.i): same function names, same argument order, numpy arrays
``[layer][y][x]``.  There is no CPU fallback: importing works anywhere, creating a model
without the built HIP library or without a GPU raises.
"""
from .api import (  # noqa: F401
    FIELDS,
    MGStats,
    MsomError,
    NODE_FIELDS,
    NodeQG,
    QG,
    init_grid,
    load_library,
    pyp2q,
    pyq2p,
    pystep_bfn,
    pystep_de,
    read_params,
    set_const,
    set_vars,
    set_vars_bfn,
    trash_vars,
    trash_vars_bfn,
)

__all__ = [
    "QG", "NodeQG", "NODE_FIELDS", "MGStats", "MsomError", "FIELDS", "load_library", "read_params", "init_grid", "set_vars",
    "set_vars_bfn", "set_const", "pystep_bfn", "pystep_de", "pyq2p", "pyp2q", "trash_vars", "trash_vars_bfn",
]
