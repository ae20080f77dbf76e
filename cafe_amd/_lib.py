"""ctypes binding of libcafe_hip.so (include/cafe_hip.h).  Fails loudly if the library is missing:
there is no CPU fallback behind these calls."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcafe_hip.so")

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p

# every symbol include/cafe_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "cafe_last_error": (ctypes.c_char_p, []),
    "cafe_version": (ctypes.c_int, []),
    "init": (None, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]),
    "batch_query": (_i32p, [_u32p, ctypes.c_int]),
    "batch_query_median": (_i32p, [_u32p, ctypes.c_int]),
    "batch_insert": (_i32p, [_u32p, ctypes.c_int]),
    "batch_insert_val": (_i32p, [_u32p, _f32p, ctypes.c_int]),
    "analyse": (ctypes.c_double, [_u32p, ctypes.c_int]),
    "analyse1": (ctypes.c_double, [_u32p, ctypes.c_int]),
    "update": (None, []),
    "save_state": (ctypes.c_int, [ctypes.c_char_p]),
    "load_state": (ctypes.c_int, [ctypes.c_char_p]),
    "cafe_global_sketch": (_vp, []),
    "cafe_sketch_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]),
    "cafe_sketch_destroy": (ctypes.c_int, [_vp]),
    "cafe_sketch_dims": (ctypes.c_int, [_vp, _i32p]),
    "cafe_sketch_scalars": (ctypes.c_int, [_vp, _f32p, _i32p, ctypes.POINTER(ctypes.c_double)]),
    "cafe_sketch_export": (ctypes.c_int, [_vp, _i32p, _f32p, _i32p, _u8p, _i32p, _f32p, _i32p, _i32p, _u8p, _u32p]),
    "cafe_sketch_import": (ctypes.c_int, [_vp, _i32p, _f32p, _i32p, _u8p, _i32p, _f32p, _i32p, _i32p, _u8p, _u32p,
                                          ctypes.c_int, ctypes.c_float, _i32p, ctypes.c_double]),
    "cafe_sketch_save": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "cafe_sketch_load": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "cafe_sketch_rebuild_free_list": (ctypes.c_int, [_vp, _vp]),
    "cafe_sketch_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int]),
    "cafe_sketch_counters": (ctypes.c_int, [_vp, _i64p]),
    "cafe_sketch_fallback_reasons": (ctypes.c_int, [_vp, _vp]),
    "cafe_sketch_clock_renorms": (ctypes.c_int, [_vp, _vp]),
    "cafe_ae_create": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double]),
    "cafe_ae_destroy": (ctypes.c_int, [_vp]),
    "cafe_ae_insert_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp]),
    "cafe_ae_query_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "cafe_ae_update": (ctypes.c_int, [_vp, _vp]),
    "cafe_ae_dims": (ctypes.c_int, [_vp, _vp]),
    "cafe_ae_export": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "cafe_ae_import": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int]),
    "cafe_ae_ref_init": (None, [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]),
    "cafe_ae_ref_batch_query": (_i32p, [_u32p, ctypes.c_int, ctypes.c_int]),
    "cafe_ae_ref_batch_insert": (_i32p, [_u32p, ctypes.c_int, ctypes.c_int]),
    "cafe_ae_ref_batch_insert_val": (_i32p, [_u32p, _f32p, ctypes.c_int, ctypes.c_int]),
    "cafe_ae_ref_update": (None, [ctypes.c_int]),
    "cafe_query_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "cafe_query_median_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "cafe_insert_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp]),
    "cafe_tables_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp,
                                          ctypes.POINTER(_vp), _i64p, _i32p]),
    "cafe_tables_destroy": (ctypes.c_int, [_vp]),
    "cafe_tables_set_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "cafe_tables_kernel_ms": (ctypes.c_int, [_vp, ctypes.c_int, _f32p]),
    "cafe_gather_que_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "cafe_migrate_que_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "cafe_forward_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "cafe_forward_hint_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "cafe_forward_hint_stats": (ctypes.c_int, [_vp, _vp]),
    "cafe_forward_ml_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "cafe_scores_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "cafe_scores_noalloc_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp]),
    "cafe_migrate_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "cafe_backward_sgd_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_float, _vp]),
    "cafe_backward_update_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_float, _vp, _vp,
                                                ctypes.c_int, _vp]),
    "cafe_forward_gid_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "cafe_migrate_gid_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp]),
    "cafe_backward_sgd_gid_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_float, _vp]),
    "cafe_backward_sgd_gid_strided_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp]),
    "cafe_backward_update_gid_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, _vp]),
    "cafe_pack_rows_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "cafe_unpack_rows_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "cafe_route_ws_words": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "cafe_route_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "cafe_split_ids_dev": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    "cafe_grad_norm_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
}

_LIB = None


class CafeError(RuntimeError):
    pass


def load():
    """Load libcafe_hip.so and bind every declared symbol.  Raises if the library was not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CafeError(f"{LIB_PATH} not found: build it with `make -C cafe_amd/csrc` "
                            "(or __graft_entry__.build()); there is no CPU fallback")
        # torch ships its own HIP runtime; load it first so that this library binds to the same runtime
        # instance (two runtimes in one process do not share devices, streams or allocations)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)           # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc, what=""):
    if rc != 0:
        raise CafeError(f"{what}: {load().cafe_last_error().decode(errors='replace')}")
