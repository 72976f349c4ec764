"""windflow_amd: MI355X-native data-stream processing (WindFlow-class capabilities).

The library itself is C++17 + HIP (``windflow_amd/include/wf``); this package
holds the Python bindings of the kernels and of the flagship pipelines, the
multi-GPU bootstrap over torch.distributed/RCCL and the build helpers.
"""
import importlib
import os

# torch first: the extension must bind to the HIP / RCCL runtime torch already
# loaded (same SONAMEs), never to a second copy
import torch  # noqa: F401

__all__ = ["load_ext", "ext_available", "include_dir"]

_ext = None


def include_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "include")


def load_ext():
    """Return the native extension module, failing loudly if it is not built."""
    global _ext
    if _ext is None:
        try:
            _ext = importlib.import_module("windflow_amd._C")
        except ImportError as e:  # pragma: no cover
            raise ImportError(
                "windflow_amd native extension is not built; run `python -m windflow_amd.build` "
                "(hipcc --offload-arch=gfx950) first: %s" % e
            )
    return _ext


def ext_available():
    try:
        load_ext()
        return True
    except ImportError:
        return False
