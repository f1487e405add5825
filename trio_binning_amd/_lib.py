"""ctypes loader for libtbk_hip.so (the C-ABI in include/tbk.h).

Mirrors the reference's loader (src/trio_binning/kmers.py:30-38: find the native library
beside the module, ``ImportError`` when it is missing) with one difference that matters:
there is exactly one backend.  If the HIP library is absent this module raises; nothing
in this package falls back to a CPU implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_NAME = "libtbk_hip.so"

TBK_OK = 0
TBK_ERR_INVALID = -1
TBK_ERR_IO = -2
TBK_ERR_FORMAT = -3
TBK_ERR_NO_DEVICE = -4
TBK_ERR_HIP = -5
TBK_ERR_NOMEM = -6
TBK_ERR_STATE = -7


class TbkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libtbk_hip: {message} (status {code})")
        self.code = code
        self.message = message


def _find():
    override = os.environ.get("TBK_LIBRARY")
    if override:
        return override
    return os.path.join(_HERE, _NAME)


_path = _find()
if not os.path.isfile(_path):
    raise ImportError(
        f"Cannot load {_NAME}: {_path} does not exist. Build it with "
        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C trio_binning_amd/csrc`. "
        "This package has no CPU fallback."
    )
# RTLD_GLOBAL: if another HIP runtime copy is loaded later into the process (PyTorch wheels
# bundle one), the system runtime this library was built against stays first in symbol
# resolution.  Load this library before importing torch.
lib = C.CDLL(_path, mode=C.RTLD_GLOBAL)

_vp = C.c_void_p
_u64 = C.c_uint64
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_dp = C.POINTER(C.c_double)


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("tbk_abi_version", C.c_int)
_sig("tbk_last_error", C.c_char_p)
_sig("tbk_device_count", C.c_int, C.POINTER(C.c_int))
_sig("tbk_device_name", C.c_int, C.c_int, C.c_char_p, C.c_size_t)
_sig("tbk_kmer_to_int", _u64, C.c_char_p, C.c_ubyte)
_sig("tbk_reverse_complement", None, C.c_char_p, C.c_char_p, C.c_ubyte)
_sig("tbk_table_create_from_file", C.c_int, C.c_char_p, C.c_int, C.POINTER(_vp))
_sig("tbk_list_parse_file", C.c_int, C.c_char_p, C.POINTER(_u64p), _u64p, C.POINTER(C.c_int))
_sig("tbk_list_free", None, _u64p)
_sig("tbk_table_create_from_keys", C.c_int, _vp, _u64, C.c_int, C.c_int, C.POINTER(_vp))
_sig("tbk_table_create_from_device_keys", C.c_int, _vp, _u64, C.c_int, C.c_int, C.POINTER(_vp))
_sig("tbk_table_destroy", None, _vp)
_sig("tbk_table_num_kmers", _u64, _vp)
_sig("tbk_table_k", C.c_int, _vp)
_sig("tbk_table_device", C.c_int, _vp)
_sig("tbk_table_distinct", C.c_int, _vp, _u64p)
_sig("tbk_table_bytes", _u64, _vp)
_sig("tbk_table_contains", C.c_int, _vp, _vp, _u64, _vp)
_sig("tbk_count_kmers_in_read", C.c_int, C.c_char_p, C.c_int64, _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("tbk_classifier_create", C.c_int, _vp, _vp, C.POINTER(_vp))
if hasattr(lib, "tbk_classifier_count_read"):
    _sig("tbk_classifier_count_read", C.c_int, _vp, C.c_char_p, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("tbk_classifier_destroy", None, _vp)
_sig("tbk_classifier_create_multi", C.c_int, _vp, _vp, C.POINTER(C.c_int), C.c_int, C.POINTER(_vp))
_sig("tbk_classifier_replicate", C.c_int, _vp, C.c_int, C.POINTER(_vp))
_sig("tbk_classifier_device", C.c_int, _vp)
_sig("tbk_classifier_stats", C.c_int, _vp, _u64p, _u64p, _u64p, _u64p)
_sig("tbk_classifier_shared_keys", C.c_int, _vp, _u64p)
_sig("tbk_classifier_layout", C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("tbk_classifier_sampling_t", C.c_int, _vp)
_sig("tbk_classifier_build_info", C.c_int, _vp, C.POINTER(C.c_int), _u64p)
if hasattr(lib, "tbk_classifier_front"):  # absent from libraries built before the front layout (tools/gpu_ab.sh compares such builds)
    _sig("tbk_classifier_front", C.c_int, _vp, C.POINTER(C.c_int), _u64p)
_sig("tbk_classify_batch", C.c_int, _vp, _vp, _vp, _u64, _vp)
_sig("tbk_stream_depth", C.c_int, _vp)
_sig("tbk_stream_submit", C.c_int, _vp, _vp, _vp, _u64, _vp, _u64p)
_sig("tbk_stream_wait", C.c_int, _vp, _u64)
_sig("tbk_packed_chunks", _u64, _u64)
_sig("tbk_pack_bases", C.c_int, _vp, _u64, _vp, _vp, _vp, _u64, _u64p)
_sig("tbk_stream_submit_packed", C.c_int, _vp, _vp, _vp, _vp, _u64, _vp, _u64, _vp, _u64p)
_sig("tbk_classifier_set_transfer", C.c_int, _vp, C.c_int)
_sig("tbk_classifier_transfer", C.c_int, _vp)
_sig("tbk_host_alloc", _vp, C.c_size_t)
_sig("tbk_host_free", None, _vp)
_sig("tbk_classify_device", C.c_int, _vp, _vp, _vp, _u64, _u64, _vp)
_sig("tbk_classifier_sync", C.c_int, _vp)
_sig("tbk_stream_submit_device", C.c_int, _vp, _vp, _vp, _u64, _u64, _vp, _u64p)
_sig("tbk_kernel_timing_enable", C.c_int, _vp, C.c_int)
_sig("tbk_kernel_timing_read", C.c_int, _vp, _u64p, _dp)
if hasattr(lib, "tbk_kernel_timing_read2"):  # (variant builds of tools/build_variant.sh may predate it)
    _sig("tbk_kernel_timing_read2", C.c_int, _vp, _u64p, _dp, _dp)
    _sig("tbk_classifier_last_passes", C.c_int, _vp, _u64p, _u64p)
if hasattr(lib, "tbk_pipeline_create"):
    _sig("tbk_device_identity", C.c_int, C.c_int, C.c_char_p, C.c_size_t)
    _sig("tbk_pipeline_create", C.c_int, _vp, _vp, C.POINTER(C.c_int), C.c_int, C.POINTER(_vp))
    _sig("tbk_pipeline_destroy", None, _vp)
    _sig("tbk_pipeline_depth", C.c_int, _vp)
    _sig("tbk_pipeline_devices", C.c_int, _vp)
    _sig("tbk_pipeline_classifier", _vp, _vp, C.c_int)
    _sig("tbk_pipeline_submit", C.c_int, _vp, _vp, _vp, _u64, _vp, _u64p)
    _sig("tbk_pipeline_submit_packed", C.c_int, _vp, _vp, _vp, _vp, _u64, _vp, _u64, _vp, _u64p)
    _sig("tbk_pipeline_wait", C.c_int, _vp, _u64, C.POINTER(C.c_int))
    _sig("tbk_pipeline_batches", C.c_int, _vp, _u64p, C.c_int)
    _sig("tbk_classify_file", C.c_int, _vp, C.c_char_p, _u64, _u64, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, _u64, _u64, _vp)
    _sig("tbk_classifier_calibrate", C.c_int, _vp, _dp)
    if hasattr(lib, "tbk_classifier_calibrate_pairs"):
        _sig("tbk_classifier_calibrate_pairs", C.c_int, _vp, C.c_int, C.c_int, _u64, _dp)
    _sig("tbk_table_origin", C.c_int, _vp)
    _sig("tbk_table_keys", C.c_int, _vp, _vp, _u64)
    _sig("tbk_fastx_set_packing", C.c_int, _vp, C.c_int)
    _sig("tbk_fastx_set_borrowing", C.c_int, _vp, C.c_int)
    _sig("tbk_fastx_batch_borrowed", C.c_int, _vp)
    _sig("tbk_classifier_entries", C.c_int, _vp, C.POINTER(C.c_int), _u64p, _u64p)
    _sig("tbk_classifier_table_id", C.c_int, _vp, _u64p, C.POINTER(C.c_int))
    _sig("tbk_device_numa_node", C.c_int, C.c_int, C.POINTER(C.c_int))
    _sig("tbk_numa_bind_to_device", C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int))
    _sig("tbk_numa_node_of_pci_", C.c_int, C.c_char_p)
    _sig("tbk_numa_node_cpus_", C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int)
    _sig("tbk_numa_bind_thread_", C.c_int, C.c_int)
    _sig("tbk_pipeline_numa", C.c_int, _vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int))
    _sig("tbk_fastx_batch_gather", C.c_int, _vp, _vp, C.c_uint64, _vp, C.c_uint64)
    _sig("tbk_fastx_batch_packed", C.c_int, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _u64p)
_sig("tbk_score_and_bin", C.c_int, _vp, _u64, _u64, _u64, _vp, _vp, _vp)
_sig("tbk_device_alloc", C.c_int, C.c_int, C.c_size_t, C.POINTER(_vp))
_sig("tbk_device_free", C.c_int, C.c_int, _vp)
_sig("tbk_memcpy_h2d", C.c_int, C.c_int, _vp, _vp, C.c_size_t)
_sig("tbk_memcpy_d2h", C.c_int, C.c_int, _vp, _vp, C.c_size_t)
_sig("tbk_device_sync", C.c_int, C.c_int)
_sig("tbk_device_mem_info", C.c_int, C.c_int, _u64p, _u64p)
_sig("tbk_synth_keys_device", C.c_int, C.c_int, _u64, _u64, _u64, C.c_int, _vp)
_sig("tbk_synth_keys_host", C.c_int, _u64, _u64, _u64, C.c_int, _vp)
_sig("tbk_synth_reads_device", C.c_int, C.c_int, _u64, _u64, _u64, C.c_uint32, _u64, _u64, _u64, C.c_int,
     C.c_int, C.c_int, _vp, _vp)
_sig("tbk_synth_hap_keys_device", C.c_int, C.c_int, _u64, _u64, C.c_uint32, C.c_int, _vp, _vp, _u64, _u64p)
_sig("tbk_synth_hap_reads_device", C.c_int, C.c_int, _u64, _u64, C.c_uint32, _u64, _u64, _u64, C.c_uint32, C.c_uint32, _vp, _vp)
if hasattr(lib, "tbk_classifier_sweep_keys"):  # (variant builds of tools/build_variant.sh may predate round 5)
    _sig("tbk_synth_lognormal_lengths", C.c_int, _u64, _u64, _u64, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, _vp)
    _sig("tbk_synth_reads_ragged_device", C.c_int, C.c_int, _u64, _u64, _u64, _vp, _u64, _u64, _u64, _u64, C.c_int, C.c_uint32, _vp)
    _sig("tbk_synth_hap_reads_ragged_device", C.c_int, C.c_int, _u64, _u64, C.c_uint32, _u64, _u64, _u64, _vp, _u64, _u64, C.c_uint32, _vp)
    _sig("tbk_table_contains_device", C.c_int, _vp, _vp, _u64, _vp)
    _sig("tbk_table_device_keys", _vp, _vp)
    _sig("tbk_synth_mutate_keys_device", C.c_int, C.c_int, _vp, _u64, _u64, C.c_int, _u64, _vp)
    _sig("tbk_sweep_expectation_device", C.c_int, _vp, _vp, _vp, _u64, _vp)
    _sig("tbk_classifier_sweep_keys", C.c_int, _vp, _vp, _u64, C.c_int, C.c_uint32, C.c_int, _vp, _u64, _u64p)
    if hasattr(lib, "tbk_classifier_verify"):
        _sig("tbk_classifier_verify", C.c_int, _vp, _vp, _vp, _u64p)
    if hasattr(lib, "tbk_classifier_verified"):
        _sig("tbk_classifier_verified", C.c_int, _vp, _u64p, C.POINTER(C.c_double))
_sig("tbk_host_threads", C.c_int)
_sig("tbk_counter_create", C.c_int, C.c_int, _u64, C.c_int, C.POINTER(_vp))
_sig("tbk_counter_destroy", None, _vp)
_sig("tbk_counter_add_batch", C.c_int, _vp, _vp, _vp, _u64)
_sig("tbk_counter_add_device", C.c_int, _vp, _vp, _vp, _u64, _u64)
_sig("tbk_counter_kernel_timing", C.c_int, _vp, _u64p, _u64p, _dp, C.c_int)
_sig("tbk_counter_histogram", C.c_int, _vp, _u64p)
_sig("tbk_counter_distinct", C.c_int, _vp, _u64p)
_sig("tbk_counter_stats", C.c_int, _vp, _u64p, _u64p, _u64p, _u64p)
_sig("tbk_counter_unique", C.c_int, _vp, _vp, C.c_uint32, C.c_uint32, C.c_char_p, _u64p)
_sig("tbk_calib_gather", C.c_int, C.c_int, _u64, C.c_int, C.c_int, C.c_int, _u64, C.c_int, _dp, _dp)
_sig("tbk_calib_atomics", C.c_int, C.c_int, _u64, C.c_int, C.c_int, _dp)
if hasattr(lib, "tbk_calib_atomics64"):
    _sig("tbk_calib_atomics64", C.c_int, C.c_int, _u64, C.c_int, C.c_int, _dp)
    _sig("tbk_counter_adds_issued", C.c_int, _vp, _u64p)
_sig("tbk_calib_stream", C.c_int, C.c_int, _u64, C.c_int, _dp)
if hasattr(lib, "tbk_calib_gather_pairs"):
    _sig("tbk_calib_gather_pairs", C.c_int, C.c_int, _u64, C.c_int, C.c_int, _u64, C.c_int, _dp)
    _sig("tbk_calib_stream_nt", C.c_int, C.c_int, _u64, C.c_int, C.c_int, C.c_int, _dp)
_sig("tbk_fastx_open", C.c_int, C.c_char_p, C.POINTER(_vp))
_sig("tbk_fastx_close", None, _vp)
_sig("tbk_fastx_batch_create", C.c_int, C.POINTER(_vp))
_sig("tbk_fastx_batch_destroy", None, _vp)
_sig("tbk_fastx_next", C.c_int, _vp, _vp, _u64, _u64)
_sig("tbk_fastx_batch_view", C.c_int, _vp, _u64p, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
     C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp))
_sig("tbk_bin_writer_open", C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_vp))
_sig("tbk_bin_writer_write", C.c_int, _vp, _vp, C.c_char_p)
_sig("tbk_bin_writer_close", C.c_int, _vp)
if hasattr(lib, "tbk_gzip_members_device"):
    _sig("tbk_bin_writer_use_device", C.c_int, _vp, C.c_int)
    _sig("tbk_bin_writer_encoder", C.c_int, _vp)
    _sig("tbk_gzip_members_device", C.c_int, C.c_int, C.c_char_p, _u64p, _u64, C.c_char_p, _u64, _u64p, _u64p)
    if hasattr(lib, "tbk_fastx_set_device"):
        _sig("tbk_fastx_set_device", C.c_int, _vp, C.c_int)
        _sig("tbk_fastx_inflates_on_device", C.c_int, _vp)
    if hasattr(lib, "tbk_bgzf_inflate_device"):
        _sig("tbk_bgzf_inflate_device", C.c_int, C.c_int, C.c_char_p, _u64, C.c_char_p, _u64, _u64p)
    if hasattr(lib, "tbk_bgzf_bench_device"):
        _sig("tbk_bgzf_bench_device", C.c_int, C.c_int, C.c_char_p, _u64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _u64p)
    if hasattr(lib, "tbk_gzip_bench_device"):
        _sig("tbk_gzip_bench_device", C.c_int, C.c_int, _vp, _u64p, _u64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _u64p)
_sig("tbk_gzip_member", C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t))
_sig("tbk_crc32_c", C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t)
_sig("tbk_format_tsv", C.c_int, _vp, C.c_char_p, _vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t))
_sig("tbk_format_float", C.c_int, C.c_double, C.c_char_p, C.c_size_t)


class tbk_options(C.Structure):
    """include/tbk.h: tbk_options (how a classifier is built, as arguments)."""
    _fields_ = [("size", C.c_uint32),
                ("short_keys", C.c_int32), ("entries", C.c_int32), ("wide_entries", C.c_int32), ("front", C.c_int32),
                ("mod_sampling", C.c_int32), ("span3", C.c_int32), ("guests", C.c_int32), ("minimizer_w", C.c_int32), ("minimizer_m", C.c_int32),
                ("two_read_kernel", C.c_int32),
                ("table_load", C.c_double), ("entry_load", C.c_double), ("wentry_load", C.c_double), ("short_load", C.c_double),
                ("clustered", C.c_double), ("behind_front", C.c_double), ("plainly_clustered", C.c_double), ("entry_min_ratio", C.c_double),
                ("memory_budget_bytes", C.c_uint64), ("table_align", C.c_uint64), ("short_line_cap", C.c_uint32), ("probe_max_blocks", C.c_int32),
                ("packed_h2d", C.c_int32), ("slice_bases", C.c_uint64), ("build_timing", C.c_int32), ("force_replica", C.c_int32),
                ("ring_streams", C.c_int32), ("copy_priority", C.c_int32), ("h2d_streams", C.c_int32), ("zero_copy", C.c_int32),
                ("full_keys", C.c_int32), ("full_load", C.c_double), ("replica_copy", C.c_int32), ("verify_build", C.c_int32)]


HAS_OPTIONS = hasattr(lib, "tbk_classifier_create_opts")  # (variant builds of tools/build_variant.sh may predate round 5)
if HAS_OPTIONS:
    _sig("tbk_options_init", None, C.POINTER(tbk_options))
    _sig("tbk_options_from_env", C.c_int, C.POINTER(tbk_options))
    _sig("tbk_classifier_create_opts", C.c_int, _vp, _vp, C.POINTER(tbk_options), C.POINTER(_vp))
    _sig("tbk_classifier_create_multi_opts", C.c_int, _vp, _vp, C.POINTER(C.c_int), C.c_int, C.POINTER(tbk_options), C.POINTER(_vp))
    _sig("tbk_pipeline_create_opts", C.c_int, _vp, _vp, C.POINTER(C.c_int), C.c_int, C.POINTER(tbk_options), C.POINTER(_vp))


def last_error() -> str:
    msg = lib.tbk_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int) -> None:
    """Raise for a non-zero tbk_status.  File problems surface as IOError (the type the
    reference raises for a missing list, kmers.py:117-118), malformed input as ValueError."""
    if status == TBK_OK:
        return
    msg = last_error()
    if status == TBK_ERR_IO:
        raise IOError(msg)
    if status in (TBK_ERR_FORMAT, TBK_ERR_INVALID):
        raise ValueError(msg)
    if status == TBK_ERR_NOMEM:
        raise MemoryError(msg)
    raise TbkError(status, msg)


def device_count() -> int:
    n = C.c_int(0)
    if lib.tbk_device_count(C.byref(n)) != TBK_OK:
        return 0
    return n.value


def warm_up() -> None:
    """Start the HIP runtime on the devices this process will use, on a thread of its own: the
    command-line drivers call this before they import numpy and parse their arguments, so that
    the runtime's start-up (device discovery, the primary context) runs beside those instead of
    after them.  Purely a head start - every entry point initialises what it needs anyway; without a
    device it returns at once."""
    import threading

    def start():
        n = C.c_int(0)
        if lib.tbk_device_count(C.byref(n)) != TBK_OK or n.value <= 0:
            return
        spec = os.environ.get("TBK_DEVICES") or os.environ.get("TBK_DEVICE") or os.environ.get("LOCAL_RANK") or ""
        wanted = [int(x) for x in spec.split(",") if x.strip().lstrip("-").isdigit()] or list(range(n.value))
        free, total = C.c_uint64(0), C.c_uint64(0)
        for device in dict.fromkeys(wanted):
            if 0 <= device < n.value:
                lib.tbk_device_mem_info(device, C.byref(free), C.byref(total))

    thread = threading.Thread(target=start, name="tbk-warm-up", daemon=True)
    thread.start()
    # a process that ends early (--help, a bad argument) lets the runtime finish starting before it
    # is torn down
    import atexit

    atexit.register(thread.join)


def device_identity(device: int = 0) -> str:
    """``<pci bus id> <uuid>`` of a device: what tells the GPUs of a multi-rank run apart."""
    buf = C.create_string_buffer(128)
    check(lib.tbk_device_identity(device, buf, 128))
    return buf.value.decode()


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    check(lib.tbk_device_name(device, buf, 256))
    return buf.value.decode()
