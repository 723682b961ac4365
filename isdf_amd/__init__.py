"""isdf_amd: MI355X-native (gfx950) training hot path of iSDF behind a C ABI.

    isdf_amd.build        hipcc build of libisdf_hip.so (include/isdf_hip.h)
    isdf_amd._ffi         ctypes binding of the C ABI
    isdf_amd.engine       tensor-level host API (sampler / inference / step / AdamW)
    isdf_amd.modules      SDFMapHIP / PositionalEncodingHIP: nn.Modules over the flat parameter buffer
    isdf_amd.hot_path     graft(trainer): the reference Trainer's hot-path methods re-bound to the kernels, in place
    isdf_amd.frame_store  keyframe store with the reference FrameData's contract (geometric-growth buffers)
    isdf_amd.dp           data-parallel protocol (one flat all-reduce message per step)
"""
__all__ = ["build", "engine", "hot_path", "modules", "frame_store", "dp"]
