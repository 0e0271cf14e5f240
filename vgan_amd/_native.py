"""ctypes binding of libvgan_gpu.so (include/vgan_gpu.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VGAN_LIB") or os.environ.get("VGAN_GPU_LIB") or os.path.join(HERE, "lib", "libvgan_gpu.so")  # override: developer A/B builds  # VGAN_LIB: a developer build (vgan_amd/build.py VGAN_BUILD_TAG)


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vgan native call failed (%d): %s" % (code, msg))
        self.code = code


VGAN_OK, VGAN_EINVAL, VGAN_ENODEV, VGAN_ENOMEM, VGAN_EIO, VGAN_ERANGE, VGAN_ESTATE = 0, -1, -2, -3, -4, -5, -6

vp = C.c_void_p


class GraphView(C.Structure):
    _fields_ = [("min_id", C.c_int64), ("max_id", C.c_int64), ("node_seq_off", vp), ("node_seq", vp),
                ("n_paths", C.c_uint32), ("mask_words", C.c_uint32), ("mask", vp), ("pangenome_base", vp),
                ("mappability", vp), ("n_mappability", C.c_uint64), ("path_names", C.c_char_p),
                ("parents_txt", C.c_char_p), ("children_txt", C.c_char_p)]


class AlnSetView(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("seq_off", vp), ("seq", vp), ("qual_off", vp), ("qual", vp), ("mapq", vp),
                ("identity", vp), ("name_off", vp), ("name", vp), ("map_off", vp), ("m_node", vp), ("m_offset", vp),
                ("m_rev", vp), ("edit_off", vp), ("e_from", vp), ("e_to", vp), ("e_seq_off", vp), ("e_seq", vp)]


class HcBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_segments", C.c_uint32), ("n_cols", C.c_uint64), ("n_qual", C.c_uint64),
                ("read_seg_off", vp), ("read_col_off", vp), ("read_qual_off", vp), ("read_algn_len", vp),
                ("read_mapq", vp), ("seg_node", vp), ("seg_start", vp), ("seg_len", vp), ("graph_seq", vp),
                ("algnseq", vp), ("qual", vp), ("on_device", C.c_int32), ("n_tileable", C.c_uint32),
                ("read_src", vp), ("packed", vp)]


class HcPackedView(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_segments", C.c_uint32), ("n_cols", C.c_uint64), ("n_qual", C.c_uint64),
                ("rhdr", vp), ("srec", vp), ("crec", vp), ("qualp", vp), ("max_read_segs", C.c_uint32),
                ("max_read_qual", C.c_uint32), ("max_read_cols", C.c_uint32), ("max_read_node_span", C.c_uint32), ("on_device", C.c_int32),
                ("read_src", vp)]


class SbSum(C.Structure):
    _fields_ = [("hi", C.c_int64), ("lo", C.c_uint64), ("nf", C.c_double)]


class FlattenStats(C.Structure):
    _fields_ = [("n_in", C.c_int64), ("n_out", C.c_int64), ("n_unmapped", C.c_int64), ("n_bad", C.c_int64),
                ("n_clamped", C.c_int64), ("n_segments", C.c_int64), ("n_cols", C.c_int64)]


class GamPipeOpts(C.Structure):
    _fields_ = [("piece_bytes", C.c_uint64), ("slots", C.c_int32), ("keep_unmapped", C.c_int32), ("mark_duplicates", C.c_int32),
                ("n_threads", C.c_int32), ("tail_bytes", C.c_uint64)]


class GamPipeStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_pieces", "compressed_bytes", "inflated_bytes", "n_messages", "n_reads", "n_duplicates",
                                          "n_device_reads", "n_host_reads", "device_bytes", "n_reanchored")] + \
               [(k, C.c_double) for k in ("ms_wall", "ms_upload", "ms_inflate", "ms_frame", "ms_parse", "ms_dedup", "ms_consume",
                                          "ms_wait_contexts")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class HcParams(C.Structure):
    _fields_ = [("background_error_prob", C.c_double), ("use_background_error_prob", C.c_int32),
                ("is_consensus_fasta", C.c_int32)]


class EukaDbView(C.Structure):
    _fields_ = [("n_clades", C.c_uint32), ("clade_id", vp), ("clade_dist", vp), ("clade_npaths", vp), ("clade_snode", vp),
                ("clade_enode", vp), ("clade_names", C.c_char_p), ("bin_off", vp), ("bin_lo", vp), ("bin_hi", vp),
                ("bin_entropy", vp)]


class DamageView(C.Structure):
    _fields_ = [("n5", C.c_uint32), ("n3", C.c_uint32), ("sub5p", vp), ("sub3p", vp)]


class EukaBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_cols", C.c_uint64), ("n_qual", C.c_uint64), ("n_maps", C.c_uint64),
                ("read_col_off", vp), ("read_qual_off", vp), ("read_map_off", vp), ("read_gseq_len", vp),
                ("read_rseq_len", vp), ("read_seq_len", vp), ("read_mapq", vp), ("read_rev", vp), ("read_src", vp),
                ("map_node", vp), ("graph_seq", vp), ("read_seq", vp), ("qual", vp), ("on_device", C.c_int32),
                ("reserved", C.c_uint32)]


class EukaFlattenStats(C.Structure):
    _fields_ = [("n_in", C.c_int64), ("n_out", C.c_int64), ("n_unmapped", C.c_int64), ("n_bad", C.c_int64)]


class EukaGamResult(C.Structure):
    _fields_ = [("n_messages", C.c_int64), ("n_mapped", C.c_int64), ("n_bad", C.c_int64), ("n_reads", C.c_int64), ("read_index", vp),
                ("read_clade", vp), ("read_pass", vp), ("read_seq_len", vp)]


class EukaParams(C.Structure):
    _fields_ = [("min_mapq", C.c_uint32), ("length_to_prof", C.c_int32)]


class EukaDetectParams(C.Structure):
    _fields_ = [("min_bins", C.c_uint32), ("min_reads", C.c_uint32), ("max_zero_bins", C.c_int32), ("entropy_threshold", C.c_double)]


class EukaReportCfg(C.Structure):
    _fields_ = [("detect", EukaDetectParams), ("length_to_prof", C.c_int32), ("run_mcmc", C.c_int32), ("iter", C.c_int32),
                ("burnin", C.c_int32), ("seed", C.c_uint64), ("out_frag", C.c_int32), ("reserved", C.c_uint32),
                ("out_group", C.c_char_p), ("out_dir", C.c_char_p)]


class EukaResults(C.Structure):
    _fields_ = [("db", vp), ("clade_count", vp), ("baseshift", vp), ("bin_cov", vp), ("n_like", vp), ("sum_log_like", vp),
                ("n_reads", C.c_int64), ("read_clade", vp), ("read_pass", vp), ("read_seq_len", vp), ("name_off", vp),
                ("names", vp)]


class EukaReadOut(C.Structure):
    _fields_ = [("clade", vp), ("in_lik", vp), ("out_lik", vp), ("like", vp), ("not_like", vp), ("pass_", vp)]


class TreeView(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("root", C.c_int32), ("parent", vp), ("dist", vp),
                ("child_off", vp), ("children", vp), ("names", C.c_char_p)]


SB_REFRESH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                            C.POINTER(C.c_uint64))
SB_MIXTURE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), C.c_double, C.POINTER(C.c_double))


SB_REFRESH_MANY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_double, C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_uint64))


class SbEngine(C.Structure):
    _fields_ = [("user", C.c_void_p), ("refresh", SB_REFRESH_FN), ("mixture", SB_MIXTURE_FN), ("refresh_many", SB_REFRESH_MANY_FN)]


class SbEstimateCfg(C.Structure):
    _fields_ = [("max_iter", C.c_uint32), ("burn", C.c_uint32), ("chains", C.c_uint32), ("n_paths", C.c_uint32), ("seed", C.c_uint64),
                ("con", C.c_double), ("freqs7", C.c_double * 7), ("run_mcmc", C.c_int32), ("quiet", C.c_int32)]


class SbBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_segments", C.c_uint32), ("n_cols", C.c_uint64), ("n_qual", C.c_uint64),
                ("read_seg_off", vp), ("read_col_off", vp), ("read_qual_off", vp), ("read_gseq_len", vp),
                ("read_rseq_len", vp), ("read_rev", vp), ("read_src", vp), ("seg_node", vp), ("seg_col", vp),
                ("seg_len", vp), ("seg_base_ix", vp), ("graph_seq", vp), ("read_seq", vp), ("qual", vp),
                ("on_device", C.c_int32), ("reserved", C.c_uint32)]


class SbFlattenStats(C.Structure):
    _fields_ = [("n_in", C.c_int64), ("n_out", C.c_int64), ("n_unmapped", C.c_int64), ("n_bad", C.c_int64)]


class SbGamResult(C.Structure):
    _fields_ = [("n_messages", C.c_int64), ("n_mapped", C.c_int64), ("n_reads", C.c_int64), ("n_bad", C.c_int64), ("n_dev_bad", C.c_int64),
                ("ms_tables", C.c_double)]


class SbParams(C.Structure):
    _fields_ = [("penalty", C.c_int32), ("reserved", C.c_int32)]


class SbSource(C.Structure):
    _fields_ = [("child", C.c_int32), ("parent", C.c_int32), ("dist", C.c_double), ("pos", C.c_double), ("theta", C.c_double)]


class SynthEukaCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_clades", C.c_uint32), ("nodes_per_clade", C.c_uint32), ("n_reads", C.c_uint64),
                ("read_len_mean", C.c_uint32), ("reserved", C.c_uint32), ("read_seed", C.c_uint64)]


class SynthGraphCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genome_len", C.c_uint32), ("n_nodes", C.c_uint32), ("n_paths", C.c_uint32)]


class SynthReadsCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_reads", C.c_uint64), ("read_len", C.c_uint32), ("indel_rate", C.c_double),
                ("softclip_rate", C.c_double), ("low_mapq_rate", C.c_double), ("errors", C.c_int32), ("first_read", C.c_uint64)]


# every symbol include/vgan_gpu.h declares: (restype, argtypes)
SYMBOLS = {
    "vgan_last_error": (C.c_char_p, []),
    "vgan_abi_version": (C.c_int, []),
    "vgan_host_release_memory": (None, [C.c_int]),
    "vgan_device_count": (C.c_int, []),
    "vgan_device_warmup": (C.c_int, [C.c_int]),
    "vgan_device_preload": (C.c_int, [C.c_int, C.c_uint]),
    "vgan_graph_load": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(vp)]),
    "vgan_graph_from_arrays": (C.c_int, [C.POINTER(GraphView), C.POINTER(vp)]),
    "vgan_graph_view_get": (C.c_int, [vp, C.POINTER(GraphView)]),
    "vgan_graph_write": (C.c_int, [vp, C.c_char_p]),
    "vgan_graph_free": (None, [vp]),
    "vgan_aln_read_gam": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "vgan_aln_parse_gam": (C.c_int, [vp, C.c_size_t, C.c_int, C.POINTER(vp)]),
    "vgan_aln_from_arrays": (C.c_int, [C.POINTER(AlnSetView), C.POINTER(vp)]),
    "vgan_aln_write_gam": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "vgan_gam_dump_json": (C.c_int, [C.c_char_p, C.c_char_p, vp]),
    "vgan_aln_view_get": (C.c_int, [vp, C.POINTER(AlnSetView)]),
    "vgan_aln_mark_duplicates": (C.c_int, [vp, vp, vp]),
    "vgan_aln_filter": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "vgan_aln_free": (None, [vp]),
    "vgan_hc_flatten": (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(vp), C.POINTER(FlattenStats)]),
    "vgan_alnparts_read_gam": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "vgan_alnparts_n_reads": (C.c_int64, [vp]),
    "vgan_alnparts_count": (C.c_int64, [vp]),
    "vgan_alnparts_first_read": (C.c_int64, [vp, C.c_int64]),
    "vgan_alnparts_mark_duplicates": (C.c_int, [vp, vp, C.POINTER(C.c_int64)]),
    "vgan_alnparts_merge": (C.c_int, [vp, C.POINTER(vp)]),
    "vgan_alnparts_free": (None, [vp]),
    "vgan_alnparts_base": (C.c_int64, [vp]),
    "vgan_gam_stream_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "vgan_gam_stream_next": (C.c_int, [vp, C.c_int64, C.POINTER(vp)]),
    "vgan_gam_stream_close": (None, [vp]),
    "vgan_gam_decode_counts": (None, [C.POINTER(C.c_int64)]),
    "vgan_host_cpus": (C.c_int, []),
    "vgan_host_cpu_account": (None, [C.POINTER(C.c_int64)]),
    "vgan_dedup_create": (C.c_int, [C.POINTER(vp)]),
    "vgan_dedup_mark": (C.c_int, [vp, vp, vp, C.POINTER(C.c_int64)]),
    "vgan_dedup_free": (None, [vp]),
    "vgan_hc_flatten_parts": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, C.c_int, C.POINTER(vp), C.POINTER(FlattenStats)]),
    "vgan_hc_flatten_masked": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, C.c_int, C.POINTER(vp), C.POINTER(FlattenStats)]),
    "vgan_hc_host_batch_get": (C.c_int, [vp, C.POINTER(HcBatch)]),
    "vgan_hc_flatten_packed": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, C.c_int, C.POINTER(vp), C.POINTER(FlattenStats)]),
    "vgan_hc_flatten_parts_packed": (C.c_int, [vp, vp, C.c_int64, C.c_int64, vp, C.c_int, C.POINTER(vp), C.POINTER(FlattenStats)]),
    "vgan_hc_host_batch_get_packed": (C.c_int, [vp, C.POINTER(HcPackedView)]),
    "vgan_hc_accumulate_packed": (C.c_int, [vp, C.POINTER(HcPackedView)]),
    "vgan_hc_packed_validate": (C.c_int, [vp, C.POINTER(HcPackedView)]),
    "vgan_hc_segment_weights_packed": (C.c_int, [vp, C.POINTER(HcPackedView), vp]),
    "vgan_hc_packed_download": (C.c_int, [vp, vp, vp, vp, vp, vp]),
    "vgan_hc_packed_view_download": (C.c_int, [C.POINTER(HcPackedView), vp, vp, vp, vp]),
    "vgan_hc_devflat_create": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "vgan_hc_devflat_run": (C.c_int, [vp, vp, vp, C.POINTER(HcPackedView), vp, C.POINTER(FlattenStats)]),
    "vgan_hc_devflat_free": (None, [vp]),
    "vgan_hc_host_batch_free": (None, [vp]),
    "vgan_reconstruct": (C.c_int, [vp, vp, C.c_int64, C.c_char_p, C.c_char_p, vp, C.c_int64, vp]),
    "vgan_hc_create": (C.c_int, [C.POINTER(GraphView), C.POINTER(HcParams), C.c_int, C.POINTER(vp)]),
    "vgan_hc_set_stream": (C.c_int, [vp, vp]),
    "vgan_hc_set_mode": (C.c_int, [vp, C.c_int]),
    "vgan_hc_reset": (C.c_int, [vp]),
    "vgan_hc_reduce_info": (C.c_int, [vp, vp]),
    "vgan_hc_reduce_last": (C.c_int, [vp, vp]),
    "vgan_hc_reduce_why": (C.c_int, [C.c_char_p, C.c_int64]),
    "vgan_gamdev_inflate_bytes": (C.c_int, [vp, C.c_uint64, vp, C.c_uint64, vp, vp]),
    "vgan_hc_devflat_run_gamdev": (C.c_int, [vp, vp, vp, C.c_int, C.c_uint32, vp, vp, vp]),
    "vgan_gamdev_mark_duplicates": (C.c_int, [vp, vp]),
    "vgan_gamdev_dup_marks": (vp, [vp]),
    "vgan_gamdev_pick": (C.c_int, [vp, vp, vp, vp]),
    "vgan_gamdev_picked": (C.c_int, [vp, vp, vp]),
    "vgan_alnparts_from_messages": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "vgan_gamdev_create": (C.c_int, [C.c_int, vp, vp]),
    "vgan_gamdev_free": (None, [vp]),
    "vgan_gamdev_drop_bytes": (C.c_int, [vp, C.c_int]),
    "vgan_gamdev_parse": (C.c_int, [vp, vp, C.c_uint64, C.c_int]),
    "vgan_gamdev_open": (C.c_int, [C.c_int, vp, vp, C.c_uint64, C.c_int, vp]),
    "vgan_hc_devflat_run_gamdev_cb": (C.c_int, [vp, vp, vp, C.c_int, C.c_uint32, vp, vp, vp, vp, vp]),
    "vgan_gamdev_sizes": (C.c_int, [vp, vp, vp]),
    "vgan_gampipe_plan": (C.c_int64, [vp, C.c_uint64, C.POINTER(GamPipeOpts), vp, vp, vp, C.c_int64]),
    "vgan_gampipe_parse_piece": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(GamPipeOpts), C.c_int64, C.POINTER(vp)]),
    "vgan_gampipe_carry_free": (None, [vp]),
    "vgan_hc_gam_start": (C.c_int, [vp, C.c_int, vp, C.c_uint64, C.POINTER(GamPipeOpts), C.POINTER(vp)]),
    "vgan_hc_gam_attach": (C.c_int, [vp, C.POINTER(vp), C.c_int, vp]),
    "vgan_hc_gam_finish": (C.c_int, [vp, C.POINTER(FlattenStats), C.POINTER(GamPipeStats)]),
    "vgan_hc_accumulate_gam_bytes": (C.c_int, [C.POINTER(vp), C.c_int, vp, vp, C.c_uint64, C.POINTER(GamPipeOpts), C.POINTER(FlattenStats),
                                              C.POINTER(GamPipeStats)]),
    "vgan_gamdev_download": (C.c_int, [vp, C.c_int, vp]),
    "vgan_hc_pack": (C.c_int, [vp, C.POINTER(HcBatch), C.POINTER(vp)]),
    "vgan_hc_packed_free": (None, [vp]),
    "vgan_hc_accumulate": (C.c_int, [vp, C.POINTER(HcBatch)]),
    "vgan_hc_segment_scalars": (C.c_int, [vp, C.POINTER(HcBatch), vp, vp]),
    "vgan_hc_batch_validate": (C.c_int, [vp, C.POINTER(HcBatch)]),
    "vgan_hc_segment_weights": (C.c_int, [vp, C.POINTER(HcBatch), vp]),
    "vgan_hc_read_loglik": (C.c_int, [vp, C.POINTER(HcBatch), vp]),
    "vgan_hc_finalize": (C.c_int, [vp, vp, vp]),
    "vgan_gbwt_load": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "vgan_gbwt_free": (None, [vp]),
    "vgan_gbwt_sequences": (C.c_int64, [vp]),
    "vgan_gbwt_bidirectional": (C.c_int, [vp]),
    "vgan_gbwt_extract": (C.c_int64, [vp, C.c_int64, vp, C.c_int64]),
    "vgan_gbwt_node_path_matrix": (C.c_int, [vp, C.c_int64, C.c_int64, vp]),
    "vgan_hc_reduce": (C.c_int, [C.POINTER(vp), C.c_int, vp, C.POINTER(C.c_int)]),
    "vgan_hc_synchronize": (C.c_int, [vp]),
    "vgan_hc_destroy": (None, [vp]),
    "vgan_hc_profile_enable": (C.c_int, [vp, C.c_int]),
    "vgan_hc_profile_read": (C.c_int, [vp, vp, vp]),
    "vgan_hc_posterior": (C.c_int, [vp, vp, C.c_char_p, C.c_char_p, C.c_int64, vp, C.c_int32]),
    "vgan_hc_argmax": (C.c_int, [vp, C.c_uint32]),
    "vgan_euka_db_load": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(vp)]),
    "vgan_euka_db_from_arrays": (C.c_int, [C.POINTER(EukaDbView), C.POINTER(vp)]),
    "vgan_euka_db_view_get": (C.c_int, [vp, C.POINTER(EukaDbView)]),
    "vgan_euka_db_free": (None, [vp]),
    "vgan_damage_from_text": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(vp)]),
    "vgan_damage_load": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(vp)]),
    "vgan_damage_view_get": (C.c_int, [vp, C.POINTER(DamageView)]),
    "vgan_damage_free": (None, [vp]),
    "vgan_euka_flatten": (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(vp), C.POINTER(EukaFlattenStats)]),
    "vgan_euka_host_batch_get": (C.c_int, [vp, C.POINTER(EukaBatch)]),
    "vgan_euka_host_batch_free": (None, [vp]),
    "vgan_euka_create": (C.c_int, [C.POINTER(EukaDbView), C.POINTER(DamageView), C.POINTER(EukaParams), C.c_int, C.POINTER(vp)]),
    "vgan_euka_set_stream": (C.c_int, [vp, vp]),
    "vgan_euka_reset": (C.c_int, [vp]),
    "vgan_euka_accumulate": (C.c_int, [vp, C.POINTER(EukaBatch), C.POINTER(EukaReadOut)]),
    "vgan_euka_finalize": (C.c_int, [vp, vp, vp, vp, vp]),
    "vgan_euka_kernel_ms": (C.c_int, [vp, vp, vp]),
    "vgan_euka_reduce": (C.c_int, [C.POINTER(vp), C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_int64)]),
    "vgan_euka_like_sums": (C.c_int, [vp, vp, vp]),
    "vgan_euka_detect": (C.c_int, [C.POINTER(EukaDbView), vp, vp, C.POINTER(EukaDetectParams), vp, vp]),
    "vgan_euka_abundance_mcmc": (C.c_int, [C.c_int32, vp, vp, vp, C.c_int32, C.c_int32, C.c_uint64, vp]),
    "vgan_euka_report": (C.c_int, [C.POINTER(EukaResults), C.POINTER(EukaReportCfg), C.c_char_p, vp, vp, vp]),
    "vgan_euka_destroy": (None, [vp]),
    "vgan_euka_synchronize": (C.c_int, [vp]),
    "vgan_euka_devflat_create": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "vgan_euka_devflat_run_gamdev": (C.c_int, [vp, vp, C.c_uint32, C.POINTER(EukaBatch), vp, C.POINTER(EukaFlattenStats)]),
    "vgan_euka_devflat_host_arrays": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "vgan_euka_devflat_free": (None, [vp]),
    "vgan_euka_batch_download": (C.c_int, [C.POINTER(EukaBatch), C.POINTER(EukaBatch)]),
    "vgan_euka_gam_start": (C.c_int, [vp, C.c_int, vp, C.c_uint64, C.POINTER(GamPipeOpts), C.POINTER(vp)]),
    "vgan_euka_gam_attach": (C.c_int, [vp, C.POINTER(vp), C.c_int, vp]),
    "vgan_euka_gam_finish": (C.c_int, [vp, C.POINTER(EukaGamResult), C.POINTER(GamPipeStats)]),
    "vgan_euka_gam_free": (None, [vp]),
    "vgan_sb_devflat_create": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "vgan_sb_devflat_expect": (C.c_int, [vp, C.c_double]),
    "vgan_sb_devflat_append_gamdev": (C.c_int, [vp, vp, C.c_uint32, vp, C.POINTER(SbFlattenStats)]),
    "vgan_sb_devflat_append_host": (C.c_int, [vp, C.POINTER(SbBatch), vp]),
    "vgan_sb_devflat_batch": (C.c_int, [vp, C.POINTER(SbBatch)]),
    "vgan_sb_devflat_free": (None, [vp]),
    "vgan_sb_batch_download": (C.c_int, [C.POINTER(SbBatch), C.POINTER(SbBatch)]),
    "vgan_sb_gam_start": (C.c_int, [vp, C.c_int, vp, C.c_uint64, C.POINTER(GamPipeOpts), C.POINTER(vp)]),
    "vgan_sb_gam_attach": (C.c_int, [vp, C.POINTER(vp), C.c_int, vp]),
    "vgan_sb_gam_finish": (C.c_int, [vp, C.POINTER(SbGamResult), C.POINTER(GamPipeStats)]),
    "vgan_sb_gam_batch": (C.c_int, [vp, C.c_int, C.POINTER(SbBatch)]),
    "vgan_sb_gam_free": (None, [vp]),
    "vgan_sb_flatten": (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(vp), C.POINTER(SbFlattenStats)]),
    "vgan_sb_host_batch_get": (C.c_int, [vp, C.POINTER(SbBatch)]),
    "vgan_sb_host_batch_free": (None, [vp]),
    "vgan_sb_create": (C.c_int, [C.POINTER(GraphView), C.POINTER(DamageView), C.POINTER(SbParams), C.c_int, C.POINTER(vp)]),
    "vgan_sb_set_stream": (C.c_int, [vp, vp]),
    "vgan_sb_precompute": (C.c_int, [vp, C.POINTER(SbBatch), vp]),
    "vgan_sb_read_tables": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "vgan_sb_loglike": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, C.c_double, vp, vp, vp, vp]),
    "vgan_sb_kernel_ms": (C.c_int, [vp, vp, vp]),
    "vgan_tree_parse": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "vgan_tree_load": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "vgan_tree_view_get": (C.c_int, [vp, C.POINTER(TreeView)]),
    "vgan_tree_free": (None, [vp]),
    "vgan_sb_engine_gpu": (C.c_int, [vp, C.POINTER(SbEngine)]),
    "vgan_sb_time_engine": (C.c_int, [vp, C.c_int]),
    "vgan_sb_resident": (C.c_int, [vp, C.c_int]),
    "vgan_sb_resident_launches": (C.c_int, [vp, vp]),
    "vgan_sb_estimate": (C.c_int, [C.POINTER(SbEngine), vp, vp, vp, C.c_uint32, C.POINTER(SbEstimateCfg), C.c_char_p]),
    "vgan_sb_best_paths": (C.c_int, [vp, vp, vp, vp]),
    "vgan_sb_mixture_loglike": (C.c_int, [vp, C.c_uint32, vp, C.c_double, vp]),
    "vgan_sb_signature_paths": (C.c_int, [vp, C.c_uint32, C.c_int64, C.c_int32, vp, vp]),
    "vgan_sb_destroy": (None, [vp]),
    "vgan_sb_sum_value": (C.c_double, [vp]),
    "vgan_sb_sum_add": (None, [vp, vp]),
    "vgan_sb_loglike_sums": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, C.c_double, vp, vp, vp]),
    "vgan_sb_mixture_sums": (C.c_int, [vp, C.c_uint32, vp, C.c_double, vp]),
    "vgan_sb_group_create": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "vgan_sb_group_free": (None, [vp]),
    "vgan_sb_engine_group": (C.c_int, [vp, C.POINTER(SbEngine)]),
    "vgan_sb_group_best_paths": (C.c_int, [vp, vp, vp]),
    "vgan_synth_euka": (C.c_int, [C.POINTER(SynthEukaCfg), vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "vgan_synth_hc_graph": (C.c_int, [C.POINTER(SynthGraphCfg), C.POINTER(vp)]),
    "vgan_synth_hc_reads": (C.c_int, [vp, C.POINTER(SynthReadsCfg), C.POINTER(vp)]),
}

ABI_VERSION = 7  # include/vgan_gpu.h: VGAN_ABI_VERSION this binding was written against

HIP_STREAM_LEGACY = 1  # hipStreamLegacy: the null stream by name (a NULL argument selects the context's own stream)


def torch_stream_ptr(device):
    """torch's current stream on `device` as a hipStream_t for the *_set_stream entry points.  torch's default stream is
    the null stream, whose handle is 0 -- which the C-ABI reads as "the context's own (non-blocking) stream"; the kernels
    would then race with torch's and RCCL's work on the null stream.  hipStreamLegacy names it explicitly."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream or HIP_STREAM_LEGACY


_lib = None


def load(path=None):
    """Load the native library; raise if it is not there (no Python/CPU fallback exists)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # PyTorch-ROCm wheels bundle their own libamdhip64; whichever HIP runtime is loaded first owns the GPUs of
    # the process.  Load torch's first so tensors and this library's kernels share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise ImportError(
            "vgan_amd: native library %s is missing -- build it with `python -m vgan_amd.build` "
            "(or __graft_entry__.build()); there is no fallback path" % p)
    L = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(L, name)  # AttributeError if the symbol is not exported
        f.restype = res
        f.argtypes = args
    if L.vgan_abi_version() != ABI_VERSION:
        raise ImportError("vgan_amd: %s speaks ABI %d, this binding ABI %d -- rebuild with `python -m vgan_amd.build`"
                          % (p, L.vgan_abi_version(), ABI_VERSION))
    if path is None:
        _lib = L
    return L


def lib():
    return load()


def check(rc):
    if rc < 0:
        raise NativeError(rc, (lib().vgan_last_error() or b"").decode(errors="replace"))
    return rc
