"""ctypes binding of libsjhip.so (the C ABI declared in include/sjhip.h).

The library is the product: if it is missing this module raises -- there is no CPU fallback.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # simdjson-go_amd/
LIB_PATH = os.environ.get("SJHIP_LIB") or os.path.join(PKG_DIR, "libsjhip.so")  # SJHIP_LIB: A/B builds

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
szp = C.POINTER(C.c_size_t)
intp = C.POINTER(C.c_int)

class StreamResult(C.Structure):
    """sjhip_stream_result (include/sjhip.h)"""
    _fields_ = [("tape", C.c_void_p), ("tape_len", C.c_size_t), ("strings", C.c_void_p), ("strings_len", C.c_size_t),
                ("message", C.c_void_p), ("message_len", C.c_size_t), ("device", C.c_int), ("records", C.c_uint64)]


# name -> (restype, argtypes); must list every symbol declared in include/sjhip.h
SYMBOLS = {
    "sjhip_supported": (C.c_int, []),
    "sjhip_device_count": (C.c_int, []),
    "sjhip_ctx_create": (C.c_void_p, [C.c_int]),
    "sjhip_ctx_destroy": (None, [C.c_void_p]),
    "sjhip_last_error": (C.c_char_p, [C.c_void_p]),
    "sjhip_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sjhip_parse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, szp, szp, szp, szp]),
    "sjhip_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sjhip_parse_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, szp, szp]),
    "sjhip_ctx_device_bytes": (C.c_size_t, [C.c_void_p]),
    "sjhip_ctx_trim": (C.c_int, [C.c_void_p]),
    "sjhip_input_block": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "sjhip_fetch_view": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "sjhip_parse_shard_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, szp, szp]),
    "sjhip_parse_shard_finish": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "sjhip_multi_create": (C.c_void_p, [C.POINTER(C.c_int), C.c_int]),
    "sjhip_multi_destroy": (None, [C.c_void_p]),
    "sjhip_multi_shards": (C.c_int, [C.c_void_p]),
    "sjhip_multi_shard_device": (C.c_int, [C.c_void_p, C.c_int]),
    "sjhip_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "sjhip_parse_nd_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, szp, szp, szp, szp]),
    "sjhip_fetch_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sjhip_parse_batch": (C.c_int, [C.c_void_p, C.c_void_p, szp, C.c_size_t, C.c_uint32, szp, szp]),
    "sjhip_parse_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, szp, szp, C.c_size_t, C.c_uint32, szp, szp]),
    "sjhip_trim_space": (None, [C.c_void_p, C.c_size_t, szp, szp]),
    "sjhip_stage1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, szp, intp]),
    "sjhip_stage1_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, szp,
                                      intp]),
    "sjhip_stage1_device_queue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int]),
    "sjhip_stage1_device_wait": (C.c_int, [C.c_void_p]),
    "sjhip_stage1_device_result": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, szp, intp]),
    "sjhip_stage1_time": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                                    C.POINTER(C.c_float)]),
    "sjhip_count_where": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, u64p]),
    "sjhip_filter_where": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, u64p, szp, szp]),
    "sjhip_fetch_filtered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sjhip_find_path": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, szp]),
    "sjhip_count_where_path": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, u64p]),
    "sjhip_project_keys": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, szp]),
    "sjhip_serialize": (C.c_int, [C.c_void_p, szp, szp, szp, szp]),
    "sjhip_serialize_ex": (C.c_int, [C.c_void_p, C.c_uint32, szp, szp, szp, szp]),
    "sjhip_deserialize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, szp, szp, szp]),
    "sjhip_fetch_message": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sjhip_fetch_serialized": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, szp]),
    "sjhip_marshal_json": (C.c_int, [C.c_void_p, szp]),
    "sjhip_fetch_marshaled": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sjhip_stream_create": (C.c_void_p, [C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_uint32]),
    "sjhip_stream_destroy": (None, [C.c_void_p]),
    "sjhip_stream_block_capacity": (C.c_size_t, [C.c_void_p]),
    "sjhip_stream_slots": (C.c_int, [C.c_void_p]),
    "sjhip_stream_in_flight": (C.c_int, [C.c_void_p]),
    "sjhip_stream_last_error": (C.c_char_p, [C.c_void_p]),
    "sjhip_stream_acquire": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), szp]),
    "sjhip_stream_grow": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]),
    "sjhip_stream_submit": (C.c_int, [C.c_void_p, C.c_size_t]),
    "sjhip_stream_cancel": (C.c_int, [C.c_void_p]),
    "sjhip_stream_submit_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "sjhip_stream_next": (C.c_int, [C.c_void_p, C.POINTER(StreamResult)]),
    "sjhip_stream_ready": (C.c_int, [C.c_void_p]),
    "sjhip_stream_set_filter": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "sjhip_stream_release": (C.c_int, [C.c_void_p]),
    "sjhip_stage1_set_variant": (C.c_int, [C.c_int]),
    "sjhip_debug_bounds_selftest": (C.c_int, []),
    "sjhip_stage1_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_uint), intp, intp]),
    "sjhip_find_odd_backslash_sequences": (C.c_int, [C.c_void_p, C.c_char_p, u64p, u64p]),
    "sjhip_find_quote_mask_and_bits": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, u64p, u64p, u64p, u64p]),
    "sjhip_find_whitespace_and_structurals": (C.c_int, [C.c_void_p, C.c_char_p, u64p, u64p]),
    "sjhip_finalize_structurals": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p,
                                             u64p]),
    "sjhip_find_newline_delimiters": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, u64p]),
    "sjhip_flatten_bits_incremental": (C.c_int, [C.c_void_p, u32p, intp, C.c_uint64, u64p, u64p]),
}

_LIB = None


class SjhipMissing(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SjhipMissing(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
