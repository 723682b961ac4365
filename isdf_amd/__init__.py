"""isdf_amd: MI355X-native (gfx950) training hot path of iSDF behind a C ABI.

    isdf_amd.build      hipcc build of libisdf_hip.so (include/isdf_hip.h)
    isdf_amd._ffi       ctypes binding of the C ABI
    isdf_amd.engine     tensor-level host API (sampler / inference / step / AdamW)
    isdf_amd.trainer    host mirror of the reference's Trainer hot-path surface
"""
__all__ = ["build", "engine", "trainer"]
