"""imfnet_amd -- MI355X-native descriptor generation for IMFNet.

Public surface (mirrors the reference's call sites):
    from imfnet_amd.model import load_model            # model/__init__.py
    from imfnet_amd.extract import extract_features    # util/misc.py:21
    import imfnet_amd.sparse as ME                      # the MinkowskiEngine symbols the path uses
"""
import os as _os
import sys as _sys


def _runtime_untouched():
    t = _sys.modules.get("torch")
    return t is None or not t.cuda.is_initialized()


# The HIP runtime's default (ROC_CPU_WAIT_FOR_SIGNAL=1) makes the CALLING THREAD wait for a stream's earlier kernels
# whenever a copy engine takes over from the compute queue: hipMemcpyAsync behind queued kernels blocks for the length of
# those kernels (measured: ~0.9 ms per call in the middle of a forward, tools/e2e_probe.py).  With 0 the dependency is
# handed to the GPU instead.  The runtime reads the variable once, when it starts, so it is set here -- at import, and
# only while the runtime has not been touched; SDMA_ASYNC tells the streaming pipeline (stream.py) whether its transfers
# may use the copy engines (hipMemcpyAsync) or must stay copy kernels (which never block, but share CUs with the forward).
# SIDE EFFECT of importing this package: ROC_CPU_WAIT_FOR_SIGNAL=0 in the process environment (unless already set).  The
# variable only DECLARES the mode; stream.py measures the effective one when it creates a pipeline (a runtime started by
# something else before this import ignores it).
# OPT-OUT: IMFNET_LEAVE_ENV=1 -- the import leaves the environment alone; the pipeline then uses the copy engines
# only if the embedding process set ROC_CPU_WAIT_FOR_SIGNAL=0 itself, and copy kernels otherwise (correct either way).
if _runtime_untouched():
    if _os.environ.get("IMFNET_LEAVE_ENV") != "1":
        _os.environ.setdefault("ROC_CPU_WAIT_FOR_SIGNAL", "0")
    SDMA_ASYNC = _os.environ.get("ROC_CPU_WAIT_FOR_SIGNAL") == "0"
else:
    SDMA_ASYNC = False

from ._lib import ImfError, LIB_PATH  # noqa: E402,F401

__version__ = "0.1.0"
