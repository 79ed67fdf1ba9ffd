"""imfnet_amd -- MI355X-native descriptor generation for IMFNet.

Public surface (mirrors the reference's call sites):
    from imfnet_amd.model import load_model            # model/__init__.py
    from imfnet_amd.extract import extract_features    # util/misc.py:21
    import imfnet_amd.sparse as ME                      # the MinkowskiEngine symbols the path uses
"""
from ._lib import ImfError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
