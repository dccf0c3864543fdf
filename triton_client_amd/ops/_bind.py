"""ctypes prototypes of ``libtca_hip.so`` (hipshm + CDNA4 kernel launchers)."""

from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_uint8, c_uint32, c_uint64, c_void_p


class PreprocImage(Structure):
    """Mirror of ``tca::PreprocImage`` (csrc/hip/kernels.h)."""

    _fields_ = [
        ("src", c_void_p), ("height", c_int), ("width", c_int), ("stride", c_int), ("channels", c_int),
        ("bgr", c_int), ("kx", c_int), ("ky", c_int), ("xb_off", c_int), ("yb_off", c_int), ("xk_off", c_int),
        ("yk_off", c_int), ("pad_", c_int),
    ]


class PreprocParams(Structure):
    """Mirror of ``tca::PreprocParams``."""

    _fields_ = [
        ("out_h", c_int), ("out_w", c_int), ("out_c", c_int), ("nhwc", c_int), ("tile_h", c_int),
        ("max_rows", c_int), ("src_pitch", c_int), ("pad_", c_int), ("image_stride", c_longlong),
        ("scale", c_float * 4),
        ("bias", c_float * 4),
    ]


# element type codes shared with the kernels (tca::DType)
DTYPE_CODE = {"FP32": 0, "FP16": 1, "BF16": 2, "INT8": 3, "UINT8": 4, "INT16": 5, "UINT16": 6, "INT32": 7,
              "UINT32": 8, "FP64": 9}
DTYPE_SIZE = {"FP32": 4, "FP16": 2, "BF16": 2, "INT8": 1, "UINT8": 1, "INT16": 2, "UINT16": 2, "INT32": 4,
              "UINT32": 4, "FP64": 8}


def install(lib):
    lib.tca_hip_last_error.restype = c_char_p
    lib.tca_hip_last_error.argtypes = []
    lib.tca_hipshm_fake_mode.restype = c_int
    lib.tca_hip_device_count.restype = c_int
    lib.tca_hip_device_count.argtypes = [POINTER(c_int)]
    lib.tca_hip_mem_info.restype = c_int
    lib.tca_hip_mem_info.argtypes = [c_int, POINTER(c_uint64), POINTER(c_uint64)]
    lib.tca_hipshm_create.restype = c_int
    lib.tca_hipshm_create.argtypes = [c_char_p, c_uint64, c_int, POINTER(c_void_p)]
    lib.tca_hipshm_info.restype = c_int
    lib.tca_hipshm_info.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_uint8), POINTER(c_int),
                                    POINTER(c_uint64), POINTER(c_int)]
    lib.tca_hipshm_write.restype = c_int
    lib.tca_hipshm_write.argtypes = [c_void_p, c_uint64, c_void_p, c_uint64]
    lib.tca_hipshm_read.restype = c_int
    lib.tca_hipshm_read.argtypes = [c_void_p, c_uint64, c_void_p, c_uint64]
    for fn in (lib.tca_hipshm_read_many, lib.tca_hipshm_write_many):
        fn.restype = c_int
        fn.argtypes = [c_int, POINTER(c_void_p), POINTER(c_uint64), POINTER(c_void_p), POINTER(c_uint64)]
    lib.tca_hipshm_copy_from.restype = c_int
    lib.tca_hipshm_copy_from.argtypes = [c_void_p, c_uint64, c_void_p, c_uint64, c_int]
    lib.tca_hipshm_copy_from_async.restype = c_int
    lib.tca_hipshm_copy_from_async.argtypes = [c_void_p, c_uint64, c_void_p, c_uint64, c_int, c_void_p]
    lib.tca_hipshm_destroy.restype = c_int
    lib.tca_hipshm_destroy.argtypes = [c_void_p]
    lib.tca_hipshm_open.restype = c_int
    lib.tca_hipshm_open.argtypes = [c_char_p, c_char_p, c_uint64, c_int, POINTER(c_void_p)]
    lib.tca_flag_error.restype = c_char_p
    lib.tca_flag_error.argtypes = []
    lib.tca_hipshm_create_ex.restype = c_int
    lib.tca_hipshm_create_ex.argtypes = [c_char_p, c_uint64, c_int, c_int, POINTER(c_void_p)]
    lib.tca_can_stream_wait_value.restype = c_int
    lib.tca_can_stream_wait_value.argtypes = [c_int, POINTER(c_int)]
    lib.tca_flag_set.restype = c_int
    lib.tca_flag_set.argtypes = [c_void_p, c_void_p, c_uint32, c_int]
    lib.tca_flag_wait.restype = c_int
    lib.tca_flag_wait.argtypes = [c_void_p, c_void_p, c_uint32, c_int, c_uint32, c_void_p]
    lib.tca_bytes_pipe_error.restype = c_char_p
    lib.tca_bytes_pipe_error.argtypes = []
    lib.tca_bytes_pipe_max_stream_bytes.restype = c_int
    lib.tca_bytes_pipe_max_stream_bytes.argtypes = []
    lib.tca_bytes_pipe_create.restype = c_int
    lib.tca_bytes_pipe_create.argtypes = [c_int, c_int, c_int, POINTER(c_void_p)]
    lib.tca_bytes_pipe_destroy.restype = None
    lib.tca_bytes_pipe_destroy.argtypes = [c_void_p]
    lib.tca_bytes_pipe_submit.restype = c_int
    lib.tca_bytes_pipe_submit.argtypes = [c_void_p, c_int, c_int, c_int, POINTER(c_uint64), POINTER(c_uint64),
                                          POINTER(c_uint64), POINTER(c_uint64), c_void_p]
    lib.tca_bytes_pipe_wait.restype = c_int
    lib.tca_bytes_pipe_wait.argtypes = [c_void_p, c_int, POINTER(c_uint32), POINTER(c_uint64)]
    lib.tca_hip_device_synchronize.restype = c_int
    lib.tca_hip_device_synchronize.argtypes = [c_int]

    lib.tca_preprocess_launch.restype = c_int
    lib.tca_preprocess_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(PreprocParams), c_int,
                                          c_int, c_void_p]
    lib.tca_bytes_pack_launch.restype = c_int
    lib.tca_bytes_pack_launch.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p, c_uint64, c_void_p]
    lib.tca_bytes_index_launch.restype = c_int
    lib.tca_bytes_index_launch.argtypes = [c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.tca_bytes_index_workspace_bytes.restype = c_uint64
    lib.tca_bytes_index_workspace_bytes.argtypes = [c_uint64]
    lib.tca_bytes_index_parallel_launch.restype = c_int
    lib.tca_bytes_index_parallel_launch.argtypes = [c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p,
                                                    c_void_p,
                                                    c_uint64, c_void_p]
    lib.tca_bytes_gather_launch.restype = c_int
    lib.tca_bytes_gather_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_void_p]
    lib.tca_exclusive_scan_u64_launch.restype = c_int
    lib.tca_exclusive_scan_u64_launch.argtypes = [c_void_p, c_uint64, c_uint32, c_void_p, c_void_p]
    lib.tca_fp32_to_bf16_launch.restype = c_int
    lib.tca_fp32_to_bf16_launch.argtypes = [c_void_p, c_void_p, c_uint64, c_int, c_void_p]
    lib.tca_bf16_to_fp32_launch.restype = c_int
    lib.tca_bf16_to_fp32_launch.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p]
    lib.tca_topk_launch.restype = c_int
    lib.tca_topk_launch.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.tca_classify_strings_launch.restype = c_int
    lib.tca_classify_strings_launch.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                                c_void_p, c_uint64, c_void_p, c_void_p]
    lib.tca_conv1x1_launch.restype = c_int
    lib.tca_conv1x1_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_void_p]
    lib.tca_conv_gemm_launch.restype = c_int
    lib.tca_conv_gemm_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.tca_avgpool_launch.restype = c_int
    lib.tca_avgpool_launch.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.tca_linear_launch.restype = c_int
    lib.tca_linear_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.tca_conv_pointwise_launch.restype = c_int
    lib.tca_conv_pointwise_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_int,
                                              c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                              c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.tca_bottleneck_rows_launch.restype = c_int
    lib.tca_bottleneck_rows_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                               c_void_p, c_void_p]
    lib.tca_conv_gemm_splitk_launch.restype = c_int
    lib.tca_conv_gemm_splitk_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                c_uint64, c_void_p, c_int, c_void_p]
    lib.tca_batched_copy_launch.restype = c_int
    lib.tca_batched_copy_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p]
    lib.tca_bias_act_launch.restype = c_int
    lib.tca_bias_act_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]
    lib.tca_bias_relu_maxpool_launch.restype = c_int
    lib.tca_bias_relu_maxpool_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.tca_stem_conv_pool_launch.restype = c_int
    lib.tca_stem_conv_pool_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    install_rt(lib)


def install_rt(lib):
    """Prototypes of csrc/hip/hip_rt.cc."""
    lib.tca_hip_rt_error.restype = c_char_p
    lib.tca_hip_set_device.restype = c_int
    lib.tca_hip_set_device.argtypes = [c_int]
    lib.tca_hip_stream_create.restype = c_int
    lib.tca_hip_stream_create.argtypes = [c_int, POINTER(c_void_p)]
    lib.tca_hip_stream_sync.restype = c_int
    lib.tca_hip_stream_sync.argtypes = [c_void_p]
    lib.tca_hip_stream_destroy.restype = c_int
    lib.tca_hip_stream_destroy.argtypes = [c_void_p]
    lib.tca_hip_host_alloc.restype = c_int
    lib.tca_hip_host_alloc.argtypes = [c_uint64, POINTER(c_void_p)]
    lib.tca_hip_host_free.restype = c_int
    lib.tca_hip_host_free.argtypes = [c_void_p]
    lib.tca_hip_malloc.restype = c_int
    lib.tca_hip_malloc.argtypes = [c_int, c_uint64, POINTER(c_void_p)]
    lib.tca_hip_free.restype = c_int
    lib.tca_hip_free.argtypes = [c_void_p]
    lib.tca_hip_memcpy_async.restype = c_int
    lib.tca_hip_memcpy_async.argtypes = [c_void_p, c_void_p, c_uint64, c_int, c_void_p]
    lib.tca_hip_memset_async.restype = c_int
    lib.tca_hip_memset_async.argtypes = [c_void_p, c_int, c_uint64, c_void_p]
    if hasattr(lib, "tca_stager_create"):
        lib.tca_stager_error.restype = c_char_p
        lib.tca_stager_error.argtypes = []
        lib.tca_stager_create.restype = c_int
        lib.tca_stager_create.argtypes = [c_int, c_uint64, c_int, c_uint64, POINTER(c_void_p)]
        lib.tca_stager_stage.restype = c_int
        lib.tca_stager_stage.argtypes = [c_void_p, c_int, POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint64),
                                         c_void_p,
                                         c_void_p]
        lib.tca_stager_stats.restype = c_int
        lib.tca_stager_stats.argtypes = [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]
        lib.tca_stager_destroy.restype = c_int
        lib.tca_stager_destroy.argtypes = [c_void_p]
