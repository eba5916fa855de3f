"""ctypes loader of libtinyknn_hip.so (C ABI: include/tinyknn_hip.h).

The HIP library is the only implementation of the hot path; there is no CPU
fallback.  Loading fails loudly if the shared object is missing, and every
compute call fails loudly if no GPU is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

ORDER_SSE, ORDER_AVX = 0, 1


class TinyKnnHipError(RuntimeError):
    pass


def lib_path():
    return os.environ.get("TINYKNN_HIP_LIB", os.path.join(_HERE, "libtinyknn_hip.so"))


_lib = None
_lib_pid = None     # the process that loaded the library (and owns every device handle made through it)

_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/tinyknn_hip.h one to one
# tk_index_set_option
OPT_SCAN_FORM, OPT_RESCORE_FORM, OPT_PLAIN_LIMIT, OPT_REPLAY_LAZY, OPT_REPLAY_COUNT, OPT_REPLAY_TWIN, OPT_TWIN_VOUCH = 1, 2, 3, 4, 5, 6, 7
OPT_PAIR_NQ, OPT_LABELS24 = 8, 9

SIGNATURES = {
    "tk_last_error": (C.c_char_p, []),
    "tk_version": (C.c_int, []),
    "tk_device_count": (C.c_int, []),
    "tk_set_device": (C.c_int, [C.c_int]),
    "tk_estimate_pq": (C.c_int, [_u64p, C.c_int64, C.c_int, _u64p, _u64p, C.c_int, C.c_int]),
    "tk_estimate_pq_batch": (C.c_int, [_u64p, C.c_int64, C.c_int, _u64p, C.c_int64, _u64p,
                                       C.c_int, C.c_int]),
    "tk_query_pq": (C.c_int, [_u64p, C.c_int64, C.c_int, C.c_int64, _u64p, _i64p, _i32p, C.c_int,
                              C.c_int, _i64p, C.c_int]),
    "tk_init_heap": (C.c_int, [_i64p, _i32p, C.c_int, C.c_int]),
    "tk_heap_insert": (C.c_int, [_i64p, _i32p, C.c_int, C.c_int64, C.c_int32]),
    "tk_heap_insert_is": (C.c_int, [_i64p, _i32p, C.c_int, C.c_int64, C.c_int32]),
    "tk_build_tables": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64,
                                  C.c_double, C.c_double, C.c_int, _u8p, C.c_void_p, _f64p]),
    "tk_knn_brute1": (C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                  C.c_int64, _i64p]),
    "tk_encode_pq": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, _u8p]),
    "tk_assign_lists": (C.c_int, [_f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int64, C.c_int, _i64p]),
    "tk_codes_upload": (C.c_void_p, [_u64p, C.c_int64, C.c_int]),
    "tk_codes_free": (None, [C.c_void_p]),
    "tk_codes_estimate": (C.c_int, [C.c_void_p, _u64p, C.c_int64, _u64p, C.c_int, C.c_int]),
    "tk_codes_estimate_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                        C.c_int, C.c_void_p]),
    "tk_codes_query": (C.c_int, [C.c_void_p, C.c_int64, _u64p, _i64p, _i32p, C.c_int, C.c_int,
                                 _i64p, C.c_int]),
    "tk_index_create": (C.c_void_p, []),
    "tk_index_destroy": (None, [C.c_void_p]),
    "tk_index_set_pq": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
    "tk_index_set_centers": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _u64p, C.c_int64]),
    "tk_index_set_lists": (C.c_int, [C.c_void_p, _i64p, _u64p, _i64p]),
    "tk_index_set_lists_shard": (C.c_int, [C.c_void_p, _i64p, _i32p, C.c_int, C.c_int, _u64p, _i64p]),
    "tk_index_shard_resident": (C.c_int, [C.c_void_p, _i32p, C.c_int, C.c_int]),
    "tk_index_shard_coarse_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p]),
    "tk_index_shard_scan_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_shard_finish_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                            C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "tk_index_shard_scan_plain_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_shard_scan_head_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_clone_shard": (C.c_void_p, [C.c_void_p, _i32p, C.c_int, C.c_int]),
    "tk_shared_stream": (C.c_void_p, [C.c_int, C.c_int]),
    "tk_index_shard_usage": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "tk_index_shard_bound_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_shard_filter_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                            C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_shard_finish_filtered_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                                     C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                                     C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_shard_plain": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tk_index_shard_plain_stats": (C.c_int, [C.c_void_p, C.c_int, _i64p]),
    "tk_index_shard_scan_first_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_shard_scan_rest_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                               C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_measure_read_bandwidth": (C.c_int, [C.c_int64, C.c_int, _f64p]),
    "tk_measure_gather_bandwidth": (C.c_int, [C.c_int64, C.c_int, C.c_int64, C.c_int, _f64p]),
    "tk_scan_exclusive_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "tk_index_replay_stats": (C.c_int, [C.c_void_p, _i64p]),
    "tk_index_twin_table": (C.c_int, [C.c_void_p, _i64p, _i32p, _i32p, _i32p]),
    "tk_index_alloc_data": (C.c_void_p, [C.c_void_p, C.c_int64, C.c_int]),
    "tk_index_synth_data": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_int,
                                      C.c_float]),
    "tk_synth_rows": (C.c_int, [_f32p, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_void_p, C.c_int,
                                C.c_float]),
    "tk_index_build_dev": (C.c_int, [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, C.c_int64, C.c_int,
                                     C.c_void_p, C.c_int, _i64p]),
    "tk_index_export_lists": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_export_centers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tk_index_read_rows": (C.c_int, [C.c_void_p, _i64p, C.c_int64, _f32p]),
    "tk_index_top_centers": (C.c_int, [C.c_void_p, _f32p, C.c_void_p, C.c_int, C.c_int64, C.c_int, _i64p]),
    "tk_index_set_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]),
    "tk_index_reserve": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "tk_index_query_batch": (C.c_int, [C.c_void_p, _f32p, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                       C.c_int, C.c_int, _i64p, _i64p, _i64p, _i32p]),
    "tk_index_query_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tk_index_set_rotation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "tk_index_prepare_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "tk_index_query_batch_raw": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                           C.c_int, _i64p]),
    "tk_host_blas_bind": (C.c_int, [C.c_char_p]),
    "tk_host_blas_bound": (C.c_int, []),
    "tk_host_threads": (C.c_int, [C.c_int]),
    "tk_prepare_queries_host": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tk_index_query_batch_dev_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                              C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "tk_index_max_sub_batch": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tk_index_pending": (C.c_int, [C.c_void_p]),
    "tk_index_input_stream": (C.c_void_p, [C.c_void_p]),
    "tk_index_info": (C.c_int, [C.c_void_p, _i64p]),
    "tk_stream_create": (C.c_void_p, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int, C.c_int]),
    "tk_stream_submit": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "tk_stream_submit_prepared": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                              C.c_void_p]),
    "tk_stream_wait": (C.c_int, [C.c_void_p, C.c_int64]),
    "tk_stream_set_probes": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "tk_stream_drain": (C.c_int, [C.c_void_p]),
    "tk_stream_prepare_seconds": (C.c_double, [C.c_void_p]),
    "tk_stream_destroy": (None, [C.c_void_p]),
    "tk_index_knn_brute": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, _i64p]),
    "tk_index_set_pipeline": (C.c_int, [C.c_void_p, C.c_int]),
    "tk_index_set_coalesce": (C.c_int, [C.c_void_p, C.c_int]),
    "tk_index_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tk_index_set_heap_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "tk_index_set_scan_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "tk_index_set_plain_scan": (C.c_int, [C.c_void_p, C.c_int]),
    "tk_index_quiesce": (C.c_int, [C.c_void_p]),
    "tk_index_plain_stats": (C.c_int, [C.c_void_p, _i64p]),
    "tk_index_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "tk_index_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "tk_index_last_profile": (C.c_int, [C.c_void_p, _f32p, _f64p, _i32p]),
}


def owns_handles():
    """False in a process forked from the one that loaded the library: its copies of the Python objects
    must not free device memory of the parent (a HIP context does not survive fork; a garbage collection
    inside e.g. a multiprocessing.Manager child used to abort with a memory fault)."""
    return _lib is not None and _lib_pid == os.getpid()


def lib():
    """The loaded library.  Raises TinyKnnHipError when it is not built."""
    global _lib, _lib_pid
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise TinyKnnHipError(
                f"{path} not found: build it with `make -C tinyknn_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). "
                "tinyknn_amd has no CPU fallback for the PQ scan path.")
        try:
            handle = C.CDLL(path)
        except OSError as e:
            raise TinyKnnHipError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        _lib_pid = os.getpid()
    return _lib


def check(rc):
    """Map a C-ABI return code to an exception.  Argument errors become
    AssertionError, the convention of the reference's Python layer
    (fast_pq.py:66,161,290,295; ivf.py:10,13,73-75)."""
    if rc >= 0:
        return rc
    msg = lib().tk_last_error().decode()
    if rc == -1:
        raise AssertionError(msg)
    raise TinyKnnHipError(msg)


def device_count():
    return lib().tk_device_count()


def ptr(a, t):
    return a.ctypes.data_as(t)
