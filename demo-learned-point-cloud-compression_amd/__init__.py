"""MI355X-native codec hot path of ikt-luh/Demo-Learned-Point-Cloud-Compression.

The directory name is not a Python identifier; import it with
    importlib.import_module("demo-learned-point-cloud-compression_amd")
or through the alias module `pcc_amd` at the repository root.

Public surface (mirrors the reference's operators):
    CompressionPipeline(settings).compress(gop)      -> (dict, sideinfo)
    DecompressionPipeline().decompress(bytes)        -> (frames, sideinfo)
"""
from ._abi import PccError, LIB_PATH  # noqa: F401


def __getattr__(name):
    # heavy imports (torch) only when the pipelines are actually used
    if name == "CompressionPipeline":
        from .codec_pipeline import CompressionPipeline
        return CompressionPipeline
    if name in ("DecompressionPipeline", "DecoderParallel"):
        from .codec_parallel import DecompressionPipeline
        return DecompressionPipeline
    if name in ("SparseTensor", "CoordSet"):
        from . import sparse
        return getattr(sparse, name)
    if name == "Runtime":
        from .runtime import Runtime
        return Runtime
    if name == "ColorModel":
        from .model import ColorModel
        return ColorModel
    if name in ("utils", "runtime", "workloads", "tiled", "metrics", "service", "capture"):   # submodules by attribute
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
