/*
 * include/vgan_gpu.h -- C-ABI of the MI355X-native per-read likelihood engine for vgan.
 *
 * The reference (grenaud/vgan) has no plugin / FFI layer: its hot path is reached through internal C++
 * calls.  Each entry point below replaces one of those call sites (paths relative to the reference's src/):
 *
 *   vgan_hc_*      Haplocart::update() loop + accumulate     HaploCart.cpp:408-421, HaploCart.h:46-50
 *                  (update_likelihood.cpp:19-53, process_mapping.cpp:4-91, get_p_obs_base.cpp:3-69)
 *   vgan_hc_posterior   Haplocart::get_posterior()           HaploCart.cpp:460, get_posterior.cpp:36-127
 *   vgan_aln_* / vgan_hc_flatten   readGAM() + reconstruct_graph_sequence() front half
 *                                                            readGAM.h:20-68, vgan_utils.h:6-79,
 *                                                            update_likelihood.cpp:28-45
 *   vgan_graph_*   readPathHandleGraph() + load_*()          readPathHandleGraph.cpp:14-37, load.cpp:6-58,283-345
 *
 * Plain pointers and sizes only; every function returns 0 on success or a negative VGAN_E* code and never
 * throws or aborts (the reference std::terminate()s on malformed reads: here they are counted and skipped).
 * One context per GPU; contexts are independent; a context is not re-entrant.
 * There is NO CPU fallback: without a HIP device every compute entry point returns VGAN_ENODEV.
 */
#ifndef VGAN_GPU_H
#define VGAN_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGAN_ABI_VERSION 7

enum {
    VGAN_OK = 0,
    VGAN_EINVAL = -1,  /* bad argument */
    VGAN_ENODEV = -2,  /* no HIP device / HIP runtime error (see vgan_last_error) */
    VGAN_ENOMEM = -3,
    VGAN_EIO = -4,     /* file could not be read / parsed */
    VGAN_ERANGE = -5,  /* value outside what the device layout can hold */
    VGAN_ESTATE = -6   /* call order violated */
};

const char *vgan_last_error(void); /* thread-local message of the last failing call */
int vgan_abi_version(void);
/* The host front end (GAM parser, flatten step, host batches) recycles its large arrays instead of returning them to the
 * system (every munmap interrupts all the cores the process runs on).  This hands the idle ones back, on n_threads threads;
 * a long-lived caller may use it between inputs, the CLI uses it on its way out. */
void vgan_host_release_memory(int n_threads);
int vgan_device_count(void);
/* Brings the HIP runtime and the device up (the first HIP call of a process costs ~0.25 s): call it on a thread of its
 * own while the graph loads.  Optional. */
int vgan_device_warmup(int device);       /* number of visible HIP devices (0 if none) */
/* (ABI 7) loads the code objects of the kernels a run will launch, now and beside each other, instead of one by one at their first
 * launches (what: VGAN_PRELOAD_* bits; call after vgan_device_warmup, on a thread of its own) */
#define VGAN_PRELOAD_GAM 1u  /* the GAM front end on the device: inflate, framing, protobuf walk */
#define VGAN_PRELOAD_HC 2u   /* HaploCart: flatten, segment kernels, mask pass */
#define VGAN_PRELOAD_EUKA 4u /* euka: read kernel, flatten */
#define VGAN_PRELOAD_SB 8u   /* soibean: analyse_GAM, refresh, flatten */
int vgan_device_preload(int device, unsigned what);

/* ------------------------------------------------------------------------------------------------
 * Graph (host side).  Replaces bdsg::ODGI + NodeInfo[] + the hcfiles sidecars.
 * The graph is read from GFA (S and P lines) or from an ODGI .og file (detected by its magic number): node ids and
 * sequences, path names in path-handle order and which paths visit each node -- the layout observed on the reference's
 * fixture test/reconstructInputSeq/target_graph.og; a file that deviates from it is rejected (VGAN_EIO).  The GBWT is not read.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_graph vgan_graph; /* opaque */

typedef struct vgan_graph_view {
    int64_t min_id, max_id;         /* node ids present: min..max (gaps allowed, empty sequence) */
    const int64_t *node_seq_off;    /* [max_id+2], indexed by NODE ID (row 0..min_id-1 empty) */
    const char *node_seq;           /* forward strand, concatenated */
    uint32_t n_paths;               /* P */
    uint32_t mask_words;            /* W = ceil(P/64) */
    const uint64_t *mask;           /* [(max_id+1)*W] bit p of row i = path_supports[i][p] (load.cpp:283-300) */
    const int32_t *pangenome_base;  /* [max_id+1] 1-based rCRS coordinate of the node (load.cpp:27-41), -1 absent */
    const double *mappability;      /* load_mappabilities() vector (load.cpp:6-24) */
    uint64_t n_mappability;
    const char *path_names;         /* P names, '\n' separated (load.cpp:43-58) */
    const char *parents_txt;        /* parents.txt / children.txt contents (load.cpp:303-345), may be "" */
    const char *children_txt;
} vgan_graph_view;

/* Load the GFA or ODGI file given and, when present in hcfiles_dir, the sidecars path_supports,
 * parsed_pangenome_mapping, mappability.tsv, graph_paths, parents.txt, children.txt (plain or .gz).
 * Missing path_supports => mask derived from the GFA P lines; missing pangenome mapping => running
 * coordinate in node-id order; missing mappability => 1.0. */
int vgan_graph_load(const char *gfa_path, const char *hcfiles_dir, vgan_graph **out);
/* Build from arrays the caller already holds (copied). */
int vgan_graph_from_arrays(const vgan_graph_view *v, vgan_graph **out);
int vgan_graph_view_get(const vgan_graph *g, vgan_graph_view *out);
int vgan_graph_write(const vgan_graph *g, const char *dir); /* graph.gfa + sidecars, the formats above */
void vgan_graph_free(vgan_graph *g);

/* ------------------------------------------------------------------------------------------------
 * Alignment set (host side) = what readGAM keeps of each vg::Alignment (readGAM.h:37-49).
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_alnset vgan_alnset; /* opaque */

typedef struct vgan_alnset_view {
    int64_t n_reads;
    const int64_t *seq_off;   /* [n_reads+1] */
    const char *seq;
    const int64_t *qual_off;  /* [n_reads+1] */
    const char *qual;         /* raw phred bytes */
    const int32_t *mapq;
    const double *identity;
    const int64_t *name_off;  /* [n_reads+1] */
    const char *name;
    const int64_t *map_off;   /* [n_reads+1] */
    const int64_t *m_node;
    const int64_t *m_offset;
    const uint8_t *m_rev;
    const int64_t *edit_off;  /* [n_mappings+1] */
    const int32_t *e_from;
    const int32_t *e_to;
    const int64_t *e_seq_off; /* [n_edits+1] */
    const char *e_seq;
} vgan_alnset_view;

/* GAM = gzip/BGZF stream of groups {varint count, count x {varint len, bytes}}, first item the tag "GAM",
 * messages = protobuf vg.Alignment (field numbers: SURVEY.md 8b).  keep_unmapped=0 drops identity==0 reads
 * exactly as readGAM.h:47 does. */
int vgan_aln_read_gam(const char *path, int keep_unmapped, vgan_alnset **out);
int vgan_aln_parse_gam(const void *bytes, size_t n, int keep_unmapped, vgan_alnset **out);
int vgan_aln_from_arrays(const vgan_alnset_view *v, vgan_alnset **out);
int vgan_aln_write_gam(const vgan_alnset *a, const char *path, int group_size);
/* `vgan haplocart -j -jf FILE` (readGAM.h:37-38, HaploCart.cpp:146-152,231): every Alignment of the GAM as one line of JSON
 * in protobuf's JSON mapping as vg's pb2json configures it -- proto field names, declaration order, absent fields left out,
 * 64-bit integers as strings, bytes as base64.  *n_alignments (or NULL): how many were written. */
int vgan_gam_dump_json(const char *gam_path, const char *json_path, int64_t *n_alignments);
int vgan_aln_view_get(const vgan_alnset *a, vgan_alnset_view *out);
/* Dup_Remover::remove_duplicates_internal (rmdup.cpp:68-110), single-end rule: is_dup[r] = 1 when an earlier read
 * has the same (node id, offset) in its first mapping.  O(n) instead of the reference's O(n^2); same marks. */
int vgan_aln_mark_duplicates(const vgan_alnset *a, uint8_t *is_dup, int64_t *n_dup);
int vgan_aln_filter(const vgan_alnset *a, const uint8_t *drop, vgan_alnset **out); /* copy without drop[r] != 0 */
void vgan_aln_free(vgan_alnset *a);

/* The same GAM kept as the slices its parser produced (input order, ~8192 reads each): duplicate marking and
 * flattening walk them in place, so a front end that only feeds the device never builds the merged set (which costs
 * as much memory again and a pass over all of it). */
typedef struct vgan_alnparts vgan_alnparts;
int vgan_alnparts_read_gam(const char *path, int keep_unmapped, vgan_alnparts **out);
int64_t vgan_alnparts_n_reads(const vgan_alnparts *p);
int64_t vgan_alnparts_count(const vgan_alnparts *p);                 /* number of slices */
int64_t vgan_alnparts_first_read(const vgan_alnparts *p, int64_t i); /* index of slice i's first read (i = count: n_reads) */
int vgan_alnparts_mark_duplicates(const vgan_alnparts *p, uint8_t *is_dup, int64_t *n_dup); /* as vgan_aln_mark_duplicates */
int vgan_alnparts_merge(vgan_alnparts *p, vgan_alnset **out);        /* consumes the slices (p stays valid, empty) */
void vgan_alnparts_free(vgan_alnparts *p);

/* The same decode pipeline handed out chunk by chunk, in input order: open() returns at once (inflate, framing and
 * parsing run behind the caller), next() blocks until the next slices holding at least min_reads reads are parsed
 * (fewer at the end; *out = NULL and VGAN_OK at the end of the stream).  Each chunk is an independent vgan_alnparts
 * (caller frees); vgan_alnparts_base = index of its first read in the whole input (read_src of its batches counts
 * from there; skip masks are indexed within the chunk). */
typedef struct vgan_gam_stream vgan_gam_stream;
int vgan_gam_stream_open(const char *path, int keep_unmapped, vgan_gam_stream **out);
int vgan_gam_stream_next(vgan_gam_stream *s, int64_t min_reads, vgan_alnparts **out);
void vgan_gam_stream_close(vgan_gam_stream *s);
/* The processors the host front end sizes its thread pools from: the smaller of the affinity mask and the cgroup CPU quota. */
int vgan_host_cpus(void);
/* Diagnostics (filled while VGAN_TIMING is set): CPU microseconds the front end's threads spent in inflate, frame + parse,
 * flatten, and the caller's part of the batch merge, summed over the process. */
void vgan_host_cpu_account(int64_t out[4]);
/* Diagnostics of the BGZF decode pipeline, summed over the process: out[0] segments decoded, out[1] segments whose own
 * framing was taken from the group it started in, out[2] from a later group, out[3] framed by the serial walk alone,
 * out[4] messages joined in a buffer of their own (longer than a segment buffer's headroom). */
void vgan_gam_decode_counts(int64_t out[5]);
int64_t vgan_alnparts_base(const vgan_alnparts *p);
/* keep-first duplicate marking across chunks (the state holds the keys seen so far; rmdup.cpp:68-110 semantics) */
typedef struct vgan_dedup vgan_dedup;
int vgan_dedup_create(vgan_dedup **out);
int vgan_dedup_mark(vgan_dedup *d, const vgan_alnparts *chunk, uint8_t *is_dup, int64_t *n_dup);
void vgan_dedup_free(vgan_dedup *d);

/* ------------------------------------------------------------------------------------------------
 * HaploCart batch: SoA produced by the front half (a1 + the slicing of update_likelihood.cpp:33-45).
 * Segment m of a read is mapping m: {node, start = position_in_read, len = |graph_seq.substr(start, mppg_sizes[m])|}.
 * The kernel compares graph_seq[col_off+start+j] with algnseq[col_off+j] (sic: from the read start,
 * update_likelihood.cpp:46) using quality qual[qual_off+start+j].
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_hc_batch {
    uint32_t n_reads;
    uint32_t n_segments;
    uint64_t n_cols;              /* bytes in graph_seq / algnseq */
    uint64_t n_qual;              /* bytes in qual */
    const uint32_t *read_seg_off; /* [n_reads+1] */
    const uint32_t *read_col_off; /* [n_reads+1] region of the read in graph_seq/algnseq (zero padded) */
    const uint32_t *read_qual_off;/* [n_reads+1] */
    const uint16_t *read_algn_len;/* [n_reads] |algnseq| (= quality window length, update_likelihood.cpp:40) */
    const uint8_t *read_mapq;     /* [n_reads] mapping quality, clamped to 99 */
    const uint32_t *seg_node;     /* [n_segments] node id */
    const uint16_t *seg_start;    /* [n_segments] first column of the mapping within its read (16 bit: see the limit below) */
    const uint16_t *seg_len;      /* [n_segments] columns it scores */
    const uint8_t *graph_seq;     /* [n_cols] ASCII incl. 'S' softclip and '-' gap marks */
    const uint8_t *algnseq;       /* [n_cols] ASCII path_string with '-' at deletions */
    const uint8_t *qual;          /* [n_qual] raw phred */
    int32_t on_device;            /* 0: host pointers (copied by accumulate); 1: device pointers (zero copy) */
    /* Reads [0, n_tileable) satisfy the tile contract below and take the LDS-tiled kernel; the others take the
     * general kernel (any length, overlapping or empty segments).  0 is always valid.  vgan_hc_flatten orders the batch
     * accordingly (tileable reads first, sorted by node id; read_src tells which read of the input each one is). */
    uint32_t n_tileable;
    const uint32_t *read_src;     /* [n_reads] index of the read in the alignment set, or NULL (not used by the device) */
    /* (ABI 3) The device-resident companion of the tileable reads in the layout the segment kernel streams, from
     * vgan_hc_pack, or NULL: vgan_hc_accumulate then runs the layout pass itself, into the context's scratch, every call. */
    const struct vgan_hc_packed *packed;
} vgan_hc_batch;
/* Batch contract: read_*_off ascending; the segments of a read ascend in seg_start.
 * Layout limit (narrower than the reference and the oracle, which have none): a read has at most 65535 alignment
 * columns, 65535 mappings and 65535 quality bytes -- seg_start / seg_len / read_algn_len are 16 bit and the general
 * kernel keeps one quality prefix per 64 bytes for 65536 of them.  vgan_hc_flatten* drops and counts reads beyond it
 * (n_bad); vgan_hc_batch_validate refuses hand-built batches beyond it.  A 16.5 kb consensus read fits four times.
 * Tile contract (reads below n_tileable): at most 1280 columns, 1280 quality bytes and 512 segments, at least one
 * segment; |algnseq| equals the length of the read's graph sequence; every segment has seg_len > 0 and the column ranges [seg_start, seg_start+seg_len)
 * of the read do not overlap.
 * Order: any.  vgan_hc_flatten* puts the tileable reads in ascending order of their lowest node id (stable), because
 * the tiled kernel keeps W[node] of a workgroup's reads in an LDS window of 448 node ids and only reaches into HBM for
 * segments outside it: a batch in another order gives the same sums, slower. */

typedef struct vgan_hc_host_batch vgan_hc_host_batch; /* opaque owner of a host-side batch */

typedef struct vgan_hc_flatten_stats {
    int64_t n_in, n_out;
    int64_t n_unmapped;   /* identity < 1e-10 (HaploCart.cpp:410) */
    int64_t n_bad;        /* reads on which the reference would std::terminate (unknown node, bad substr, ...), and
                           * reads beyond the batch layout's 16-bit limits (more than 65535 columns, mappings or quality bytes) */
    int64_t n_clamped;    /* mapq >= 100 clamped (reference reads out of bounds) */
    int64_t n_segments;
    int64_t n_cols;
} vgan_hc_flatten_stats;

/* reads [r0, r1) of the alignment set -> batch.  n_threads <= 0: all hardware threads. */
int vgan_hc_flatten(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, int n_threads,
                    vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats);
/* the same, leaving out the reads with skip[r] != 0 (skip is indexed like the alignment set, e.g. the marks of
 * vgan_aln_mark_duplicates; NULL = none): duplicate removal without rebuilding the alignment set */
int vgan_hc_flatten_masked(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, const uint8_t *skip,
                           int n_threads, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats);
/* slices [part0, part1) of a sliced set; skip (or NULL) is indexed by read within the set, read_src = base + that index */
int vgan_hc_flatten_parts(const vgan_graph *g, const vgan_alnparts *p, int64_t part0, int64_t part1, const uint8_t *skip,
                          int n_threads, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats);
int vgan_hc_host_batch_get(const vgan_hc_host_batch *b, vgan_hc_batch *out);
void vgan_hc_host_batch_free(vgan_hc_host_batch *b);

/* (ABI 4) The packed batch: the reads that satisfy the tile contract, in the layout the segment kernel streams -- what the
 * front half hands to the device for every read the kernel can take whole (HaploCart touches each read once,
 * HaploCart.cpp:408-421: the layout is written once, by the flatten step, and nothing on the device re-arranges it).
 * Bytes are moved and nothing else: no comparison, clamp or table lookup happens before the kernel.
 *   rhdr   uint32 [4 * (n_reads + 1)]  per read {first segment, first quality byte, first column, |algnseq| | mapq << 16};
 *                                      entry n_reads holds the three end offsets
 *   srec   uint32 [n_segments]         per mapping VGAN_HC_SREC(node id, seg_start, read index): node id in bits 0-17,
 *                                      seg_start in bits 18-28, the read's index & 7 in bits 29-31 ((ABI 5) one word a mapping;
 *                                      a graph with node ids beyond VGAN_HC_SREC_MAX_NODE hands every read over in the SoA form)
 *   crec   uint32 [n_cols]             per alignment column, at the column's own position: byte 0 graph_seq[c], byte 1
 *                                      algnseq[c - seg_start] (the read bases are taken from the READ start,
 *                                      update_likelihood.cpp:46), byte 2 qual[c] (0 past the quality string; (ABI 5) on EVERY
 *                                      column, scored or not: the kernel takes the quality prefix sums of
 *                                      get_log_lik_if_unsupported, process_mapping.cpp:4-24, from the column records it holds
 *                                      anyway), byte 3 = VGAN_HC_CREC_HEAD >> 24 on the first column of a mapping and 0
 *                                      elsewhere; bytes 0 and 1 of a column no mapping scores are 0
 *   qualp  uint8 [n_qual + 32]         the quality strings, followed by 32 zero bytes (aligned 8-byte words are read whole).
 *                                      Read by the per-segment kernel forms only: vgan_hc_accumulate_packed in the node-weights
 *                                      mode neither uploads nor reads it when the kernel that takes the batch finds the quality
 *                                      bytes in the column records, and accepts NULL here then (VGAN_EINVAL otherwise)
 * (ABI 5) The tile contract also asks for a quality string no longer than the read's columns (|quality| <= |algnseq|: the
 * parser takes the two lengths independently); a read that breaks it stays with the SoA batch.
 * Reads in ascending order of their lowest node id, those of mapping quality VGAN_HC_MAPQ_MAJOR first and the others behind
 * them (any order gives the same sums, slower: the kernel keeps W[node] of a wave's reads in a window over neighbouring node
 * ids, and takes whole tiles of reads that share the one mapping quality through a table of column terms).  The per-read
 * maxima select the kernel variant.  on_device: the four arrays are device pointers (a batch resident in HBM, zero copy). */
#define VGAN_HC_SREC_MAX_NODE 0x3FFFFu
#define VGAN_HC_SREC(node, start, read) ((uint32_t)(node) | (uint32_t)(start) << 18 | ((uint32_t)(read) & 7u) << 29)
#define VGAN_HC_CREC_HEAD 0x04000000u /* (ABI 5; was bit 31) a column's running count of these IS its mapping's index * 4 */
#define VGAN_HC_MAPQ_MAJOR 60         /* vg giraffe's cap: the mapping quality of (nearly) every uniquely placed read */
typedef struct vgan_hc_packed_view {
    uint32_t n_reads;
    uint32_t n_segments;
    uint64_t n_cols;
    uint64_t n_qual;
    const uint32_t *rhdr;
    const uint32_t *srec;
    const uint32_t *crec;
    const uint8_t *qualp;
    uint32_t max_read_segs, max_read_qual, max_read_cols; /* over the reads: <= 512 / 1280 / 1280 (the tile contract) */
    uint32_t max_read_node_span; /* largest (highest - lowest node id) of a read, or 0 when not known: the kernel that adds columns
                                  * straight into its window of W[node] is taken when the reads fit that window */
    int32_t on_device;
    const uint32_t *read_src; /* [n_reads] index of each read in the alignment set (host; not used by the device), or NULL */
} vgan_hc_packed_view;
/* As vgan_hc_flatten_masked / vgan_hc_flatten_parts, but the reads that satisfy the tile contract leave in the packed layout
 * (vgan_hc_host_batch_get_packed) and vgan_hc_host_batch_get returns the OTHER reads alone, as a batch of its own
 * (n_tileable = 0, offsets from 0).  A job hands both to the context: vgan_hc_accumulate_packed + vgan_hc_accumulate. */
int vgan_hc_flatten_packed(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, const uint8_t *skip, int n_threads,
                           vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats);
int vgan_hc_flatten_parts_packed(const vgan_graph *g, const vgan_alnparts *p, int64_t part0, int64_t part1, const uint8_t *skip,
                                 int n_threads, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats);
int vgan_hc_host_batch_get_packed(const vgan_hc_host_batch *b, vgan_hc_packed_view *out);
/* a1 on its own, for the reconstruction KATs: strings are NUL terminated into caller buffers of cap bytes. */
int vgan_reconstruct(const vgan_graph *g, const vgan_alnset *a, int64_t r, char *graph_seq, char *read_seq,
                     int32_t *mppg_sizes, int64_t cap, int64_t *lens /* [3] */);

/* ------------------------------------------------------------------------------------------------
 * GBWT (the reference loads <dbprefix>.gbwt beside the ODGI graph and walks every path with gbwt->extract(),
 * readOG_Euka.h:36-74).  Reads the file version `vg gbwt -o` of the reference's pinned vg writes (version 4, sdsl
 * serialisation, with or without vg's "GBWT" type-tag framing); layout pinned on test/reconstructInputSeq/target_graph.gbwt
 * against the P lines of target_graph.gfa, anything else is refused.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_gbwt vgan_gbwt; /* opaque */
int vgan_gbwt_load(const char *path, vgan_gbwt **out);
void vgan_gbwt_free(vgan_gbwt *g);
int64_t vgan_gbwt_sequences(const vgan_gbwt *g);   /* threads; a bidirectional index holds 2 per path (forward 2k, reverse 2k+1) */
int vgan_gbwt_bidirectional(const vgan_gbwt *g);
/* gbwt::GBWT::extract(sequence): the thread's nodes in GBWT encoding (2 * node id + is_reverse).  Returns the length
 * (writes at most cap entries; an id beyond the index extracts nothing, as in gbwt) or a negative error. */
int64_t vgan_gbwt_extract(const vgan_gbwt *g, int64_t sequence, uint64_t *nodes, int64_t cap);
/* The node x path matrix as readOG_Euka.h:55-73 fills it, quirks kept: extract(path_id) for path_id < n_paths (GBWT
 * sequence ids, not path ids) and the ENCODED node numbers used as node ids (row = encoding - 1).  matrix: uint8
 * [n_nodes][n_paths], row major.  The reference never reads the result beyond an emptiness check (the assignment into
 * NodeInfo::pathsgo at :98 is commented out; soibean takes path membership from the ODGI's own paths, soibean.cpp:476-491). */
int vgan_gbwt_node_path_matrix(const vgan_gbwt *g, int64_t n_nodes, int64_t n_paths, uint8_t *matrix);

/* ------------------------------------------------------------------------------------------------
 * HaploCart device context.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_hc_ctx vgan_hc_ctx; /* opaque */

typedef struct vgan_hc_params {
    double background_error_prob;      /* -e (HaploCart.cpp:70,105-113) */
    int32_t use_background_error_prob; /* set for -f consensus FASTA (HaploCart.cpp:397-400) */
    int32_t is_consensus_fasta;
} vgan_hc_params;

enum {
    /* ll[p] accumulated through per-node weights: W[node] += S_m - U_m, then one bitmask pass
     * (SURVEY.md 8a "K1 algebra"); same sums, O(R*M) instead of O(R*M*P). Default. */
    VGAN_HC_MODE_NODE_WEIGHTS = 0,
    /* every segment streams its 8*ceil(P/64)-byte mask row and updates all P accumulators
     * (the reference's per-read x per-path loop, process_mapping.cpp:54-88). */
    VGAN_HC_MODE_PER_READ = 1,
    /* as PER_READ but without skipping mask tiles that are all-supported: every row's words are applied
     * (data-independent cost; the figure the dense roofline is quoted on). */
    VGAN_HC_MODE_PER_READ_DENSE = 2
};

int vgan_hc_create(const vgan_graph_view *graph, const vgan_hc_params *params, int device, vgan_hc_ctx **out);
/* hip_stream: a hipStream_t; NULL = the context's own (non-blocking) stream.  The null stream has the handle 0 too: to
 * run on it -- e.g. beside PyTorch / RCCL work queued on torch's default stream -- pass hipStreamLegacy ((hipStream_t)1).
 * The same holds for vgan_euka_set_stream and vgan_sb_set_stream. */
int vgan_hc_set_stream(vgan_hc_ctx *c, void *hip_stream);
int vgan_hc_set_mode(vgan_hc_ctx *c, int mode);
int vgan_hc_reset(vgan_hc_ctx *c);                         /* zero the accumulators */
/* Full check of a batch held in host memory against the contracts above and the context's graph (offsets ascending
 * and within the arrays, node ids known, segments inside their read's columns, the tile contract for the first
 * n_tileable reads).  vgan_hc_accumulate only checks for null arrays: batches from vgan_hc_flatten* hold by
 * construction, hand-built ones should be validated once (O(reads + segments) on the host). */
int vgan_hc_batch_validate(const vgan_hc_ctx *c, const vgan_hc_batch *batch);
/* The layout pass on its own, for a batch that stays resident in HBM and is accumulated more than once: the tileable reads
 * go, once, into the layout the segment kernel streams -- a 32-bit record per alignment column {graph byte, read byte as
 * update_likelihood.cpp:46 pairs them, quality byte, first-column-of-a-mapping bit}, an 8-byte record per mapping, a
 * 16-byte header per read.  Bytes are moved, nothing is compared, clamped or looked up.  The result belongs to the
 * context's device, is independent of the batch's own arrays afterwards (the reads beyond n_tileable still use them) and is
 * handed back through vgan_hc_batch.packed.  Synchronises the context's stream. */
typedef struct vgan_hc_packed vgan_hc_packed; /* opaque */
int vgan_hc_pack(vgan_hc_ctx *c, const vgan_hc_batch *batch, vgan_hc_packed **out);
void vgan_hc_packed_free(vgan_hc_packed *p);
/* What the layout pass wrote, copied to host arrays sized as in vgan_hc_packed_view (test aid: the host flatten's packed
 * layout is held against it word for word).  n[4] receives reads, segments, columns, quality bytes; any array may be NULL. */
int vgan_hc_packed_download(const vgan_hc_packed *p, uint64_t n[4], uint32_t *rhdr, uint32_t *srec, uint32_t *crec, uint8_t *qualp);
/* the arrays of a packed view (host or device: the current device must be the view's) copied to host arrays sized as the view
 * says (test aid; any array may be NULL) */
int vgan_hc_packed_view_download(const vgan_hc_packed_view *v, uint32_t *rhdr, uint32_t *srec, uint32_t *crec, uint8_t *qualp);
/* Asynchronous on the context's stream: adds the batch's reads into the device accumulators. */
int vgan_hc_accumulate(vgan_hc_ctx *c, const vgan_hc_batch *batch);
/* (ABI 4) The same for a packed batch: host arrays are copied to the device as they are (one copy of the batch in HBM, no
 * layout pass), device arrays are read in place.  All three modes. */
int vgan_hc_accumulate_packed(vgan_hc_ctx *c, const vgan_hc_packed_view *batch);
/* (ABI 5) The GAM front end on the device (csrc/gam_kernels.hip; reference: src/readGAM.h:20-68).  Test / developer entry of its
 * first stage: a BGZF file's bytes are inflated on the device (a lane per BGZF member) and copied back; out = NULL asks for the
 * inflated size alone.  kernel_ms (or NULL): the inflate kernel's device time. */
int vgan_gamdev_inflate_bytes(const void *bytes, uint64_t n, void *out, uint64_t out_cap, uint64_t *out_size, double *kernel_ms);
/* (ABI 5) The whole front end: a BGZF GAM file's bytes -> the parser's arrays on the device, as kernels: inflate (a lane per BGZF
 * member), framing of libvgio's groups (a lane per 1 MiB segment, from the first group tag "GAM" it finds to the next segment's: the
 * walks must meet -- a segment whose first tag-like bytes the walk in front does not arrive at takes the next ones --, or the call fails
 * with VGAN_EIO and the caller takes the host pipeline), protobuf wire walk of vg.Alignment (a lane
 * per message: sizes, exclusive sums, fill).  What it leaves on the device is, bit for bit, what vgan_gam_stream's parser and the
 * narrowing of vgan_hc_devflat_run make of the same file: 32-bit offsets, node ids, mapping offsets, edit lengths (-1: not a match
 * or substitution), quality and substitution bytes, the identity as HaploCart.cpp:410's one bit, and the first mapping's
 * (node id, offset) per read for the duplicate marks (src/rmdup.cpp).  keep_unmapped = 0 drops identity == 0 (readGAM.h:47).
 * hip_stream NULL: a stream of the object's own. */
typedef struct vgan_gamdev vgan_gamdev;
int vgan_gamdev_create(int device, void *hip_stream, vgan_gamdev **out);
void vgan_gamdev_free(vgan_gamdev *g);
/* Gives device memory back before the object goes: what = 1 the file's bytes (done with when a parse returns), what = 2 the inflated
 * bytes as well (after which vgan_gamdev_pick fails with VGAN_EINVAL until the next parse; the parsed arrays stay).  A caller that
 * frees on a thread of its own beside later kernels shortens the process's end: the driver takes 44 GB apart in ~0.2 s. */
int vgan_gamdev_drop_bytes(vgan_gamdev *g, int what);
int vgan_gamdev_parse(vgan_gamdev *g, const void *bytes, uint64_t n, int keep_unmapped);
/* sizes[8]: inflated bytes, messages, reads, mappings, edits, edit-sequence bytes, quality bytes, and (test aid) the tag-like bytes the
 * framing of the object's parses took for a group's tag and gave up again; ms[4]: upload, inflate, framing, parsing (wall, synchronous) */
int vgan_gamdev_sizes(const vgan_gamdev *g, uint64_t sizes[8], double ms[4]);
/* Duplicate marks of the last parse's reads on the device (src/rmdup.cpp's single-end rule, keep-first by the first mapping's (node
 * id, offset): two stable radix sorts + a mark pass); vgan_gamdev_dup_marks: the device array (uint8 per read), for
 * vgan_hc_devflat_run_gamdev(skip = it, skip_on_device = 1). */
int vgan_gamdev_mark_duplicates(vgan_gamdev *g, int64_t *n_dup);
/* vgan_gamdev_create + vgan_gamdev_parse in one call, the host's walk over the BGZF member headers made BEFORE the first HIP call: a
 * thread that calls this while another thread brings the HIP runtime up (~0.25 s in a fresh process) has the index ready when the
 * runtime is.  *out is null on failure. */
int vgan_gamdev_open(int device, void *hip_stream, const void *bytes, uint64_t n, int keep_unmapped, vgan_gamdev **out);
const uint8_t *vgan_gamdev_dup_marks(const vgan_gamdev *g);
/* The messages of the reads read_mask names (host, uint8 per read of the last parse: the device flatten's host_mask), gathered on the
 * device; vgan_gamdev_picked copies them down (offsets [n_msgs + 1], bytes [n_bytes]); vgan_alnparts_from_messages parses them
 * (slices of them on n_threads threads; <= 0: all hardware threads). */
int vgan_gamdev_pick(vgan_gamdev *g, const uint8_t *read_mask, uint64_t *n_msgs, uint64_t *n_bytes);
int vgan_gamdev_picked(const vgan_gamdev *g, uint64_t *offsets, uint8_t *bytes);
int vgan_alnparts_from_messages(const uint8_t *bytes, const uint64_t *offsets, int64_t n, int keep_unmapped, int n_threads, vgan_alnparts **out);
/* test aid: array `which` of the last parse copied to the host (the list is beside the definition, csrc/gam_kernels.hip) */
int vgan_gamdev_download(const vgan_gamdev *g, int which, void *dst);
/* (ABI 4) a1 on the device: reconstruct_graph_sequence (vgan_utils.h:6-79), the slicing of update_likelihood.cpp:28-45 and the
 * packed layout in one pass over a chunk of the parser's arrays, for the reads whose edits are all matches or substitutions
 * on known nodes and which satisfy the tile contract -- what it writes is vgan_hc_flatten_parts_packed's packed batch of those
 * reads, word for word, resident in HBM.  Every other read (indels, soft clips, long reads, reads the reference would
 * terminate on) is left to the host: host_mask[r] = 1 (r indexes the chunk's reads; caller array of vgan_alnparts_n_reads
 * bytes), to be flattened with vgan_hc_flatten_parts_packed(skip = the complement of host_mask, joined with the caller's own
 * skip marks).  skip (or NULL): reads to leave out altogether, indexed like host_mask.  out: a device view (on_device = 1)
 * valid until the next run on this object; out->read_src is a host array.  stats: n_in / n_out / n_unmapped / n_clamped /
 * n_segments / n_cols of the reads taken here.  Synchronises the context's stream. */
typedef struct vgan_hc_devflat vgan_hc_devflat;
int vgan_hc_devflat_create(vgan_hc_ctx *c, const vgan_graph *graph, vgan_hc_devflat **out);
int vgan_hc_devflat_run(vgan_hc_devflat *f, const vgan_alnparts *chunk, const uint8_t *skip, vgan_hc_packed_view *out,
                        uint8_t *host_mask, vgan_hc_flatten_stats *stats);
void vgan_hc_devflat_free(vgan_hc_devflat *f);
/* (ABI 5) The same over the arrays a vgan_gamdev_parse left on the device (declared below): nothing crosses the link but the duplicate
 * marks (skip: per read of the parse, host memory or -- skip_on_device -- device memory; NULL: none) and the mask of the reads left to
 * the host.  base: index of the parse's first read in the whole input (read_src = base + index). */
struct vgan_gamdev;
int vgan_hc_devflat_run_gamdev(vgan_hc_devflat *f, const struct vgan_gamdev *gd, const uint8_t *skip, int skip_on_device, uint32_t base,
                               vgan_hc_packed_view *out, uint8_t *host_mask, vgan_hc_flatten_stats *stats);
/* The same, with mask_ready(user) called (on the calling thread) as soon as host_mask is final -- before the offsets and the write pass:
 * the caller's work on the reads left to the host (vgan_gamdev_pick ...) can run beside them, on a thread the callback starts. */
int vgan_hc_devflat_run_gamdev_cb(vgan_hc_devflat *f, const struct vgan_gamdev *gd, const uint8_t *skip, int skip_on_device, uint32_t base,
                                  vgan_hc_packed_view *out, uint8_t *host_mask, vgan_hc_flatten_stats *stats, void (*mask_ready)(void *user),
                                  void *user);
/* (ABI 6) The GAM front end on the device as a PIPELINE over the file's pieces (csrc/gam_pipe.hip; reference: src/readGAM.h:20-68 feeding
 * HaploCart.cpp:383-421, readGAM_Euka.h:581, getLCAfromGAM.h:31-45).  The BGZF file is cut at member boundaries into pieces of at most
 * piece_bytes compressed bytes; every piece goes upload -> inflate -> framing -> protobuf walk -> (duplicate marks) -> device flatten ->
 * likelihood kernels on one of `slots` fixed sets of device buffers of its lane, each set driven by a host thread and a stream of its own:
 * piece k + 1 is copied and inflated while piece k is framed and parsed and piece k - 1 is flattened and accumulated.  What crosses from a
 * piece to the next: the framing walk's state and the bytes of the item the piece's end cut (a few hundred bytes: they go in front of the
 * next piece's inflated bytes), the index of its first read, and -- with duplicate marks -- the (node id, offset) keys seen so far
 * (src/rmdup.cpp keeps the FIRST read of a key in input order).  With several lanes (one per context: --gpus LIST) piece i goes to lane
 * i mod n: every device inflates and parses its own pieces, and no data-path collective is needed before the final reduce.  Device
 * memory: slots x ~(1 + 2.1 x compression ratio) x piece_bytes + one flattened piece per lane, whatever the file's size.
 * Every array a piece's parse leaves is, bit for bit, the host parser's for the same reads (tests/test_gampipe_gpu.py). */
typedef struct vgan_gampipe_opts {
    uint64_t piece_bytes;     /* compressed bytes per piece at most; 0: 1/24 of a lane's share of the file within 32..192 MB (VGAN_GAMPIPE_PIECE overrides) */
    int32_t slots;            /* pieces in flight per lane; 0: 3 (VGAN_GAMPIPE_SLOTS overrides) */
    int32_t keep_unmapped;    /* 0: identity == 0 is dropped (readGAM.h:47) */
    int32_t mark_duplicates;  /* src/rmdup.cpp's single-end rule */
    int32_t n_threads;        /* host threads for the reads the device flatten leaves to the host; <= 0: what the process may use */
    uint64_t tail_bytes;      /* room for the item a piece's end cuts; 0: 8 MB (an item longer than this: VGAN_ERANGE) */
} vgan_gampipe_opts;
typedef struct vgan_gampipe_stats {
    uint64_t n_pieces, compressed_bytes, inflated_bytes, n_messages, n_reads, n_duplicates;
    uint64_t n_device_reads, n_host_reads; /* flattened on the device / handed back to the host as their messages */
    uint64_t device_bytes;                 /* device memory the front end held at its peak, summed over lanes (the contexts' own not counted) */
    uint64_t n_reanchored;                 /* (test aid) tag-like bytes taken for a group's tag and given up again */
    double ms_wall;                        /* start -> finish */
    double ms_upload, ms_inflate, ms_frame, ms_parse, ms_dedup, ms_consume; /* summed over pieces (they overlap: not a partition of ms_wall) */
    double ms_wait_contexts;               /* what the first pieces waited for vgan_hc_gam_attach */
} vgan_gampipe_stats;
/* The pieces the options cut a BGZF buffer into (host only; test aid): returns their number, or < 0 when the buffer is not BGZF;
 * first_member / n_members (or NULL): [n] of them, capacity cap. */
int64_t vgan_gampipe_plan(const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, uint64_t *piece_in_off, uint64_t *piece_in_bytes,
                          uint64_t *piece_out_bytes, int64_t cap);
/* (test aid) One piece of the plan through one vgan_gamdev object, as the pipeline's slot threads do it: upload + inflate, framing from
 * *carry (an opaque state, NULL-initialised by the caller for piece 0, freed by vgan_gampipe_carry_free), protobuf walk.  The object then
 * holds the piece's arrays (vgan_gamdev_download, vgan_gamdev_sizes). */
typedef struct vgan_gampipe_carry vgan_gampipe_carry;
int vgan_gampipe_parse_piece(vgan_gamdev *g, const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, int64_t piece, vgan_gampipe_carry **carry);
void vgan_gampipe_carry_free(vgan_gampipe_carry *c);
/* HaploCart over a BGZF GAM's bytes: everything from the file's bytes to W[node] / Stot of the contexts (one lane per context, which
 * may share devices).  start: host-only index of the first piece, then threads; upload, inflate, framing and parse begin at once and do
 * not need the contexts -- attach hands them over when they are ready (the flatten and the kernels start then); finish waits for the
 * last piece, reports and frees.  A failure (not BGZF, a member that does not inflate, a stream that cannot be framed, no memory) is
 * reported by finish: the contexts then hold a partial sum -- vgan_hc_reset them and take the host pipeline.  `bytes` must stay valid
 * until finish returns.  vgan_hc_accumulate_gam_bytes: the three in one call. */
typedef struct vgan_hc_gamrun vgan_hc_gamrun;
int vgan_hc_gam_start(const int *devices, int n_lanes, const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, vgan_hc_gamrun **out);
int vgan_hc_gam_attach(vgan_hc_gamrun *r, vgan_hc_ctx *const *ctxs, int n_ctx, const vgan_graph *graph);
int vgan_hc_gam_finish(vgan_hc_gamrun *r, vgan_hc_flatten_stats *stats, vgan_gampipe_stats *pstats);
int vgan_hc_accumulate_gam_bytes(vgan_hc_ctx *const *ctxs, int n_ctx, const vgan_graph *graph, const void *bytes, uint64_t n,
                                 const vgan_gampipe_opts *opts, vgan_hc_flatten_stats *stats, vgan_gampipe_stats *pstats);
/* Host check of a packed batch against the layout above and the context's graph (offsets, node ids, head bits, maxima). */
int vgan_hc_packed_validate(const vgan_hc_ctx *c, const vgan_hc_packed_view *batch);
/* D_m = S_m - U_m per segment of a packed batch (test / debug aid). Host output [n_segments]. */
int vgan_hc_segment_weights_packed(vgan_hc_ctx *c, const vgan_hc_packed_view *batch, double *D);
/* Per-segment scalars of a batch (test / debug aid): S_m, U_m as the kernel computes them. Host outputs. */
int vgan_hc_segment_scalars(vgan_hc_ctx *c, const vgan_hc_batch *batch, double *S, double *U);
/* D_m = S_m - U_m per segment as vgan_hc_accumulate computes it, i.e. through the LDS-tiled kernel for the batch's
 * tileable reads (test / debug aid). Host output [n_segments]. */
int vgan_hc_segment_weights(vgan_hc_ctx *c, const vgan_hc_batch *batch, double *D);
/* Per-read log-likelihood vectors (the value Haplocart::update returns), [n_reads*n_paths] host doubles. */
int vgan_hc_read_loglik(vgan_hc_ctx *c, const vgan_hc_batch *batch, double *out);
/* final_vec[P] (HaploCart.cpp:420) of everything accumulated so far. d_out: device double[P], filled
 * asynchronously on the stream (hand it to RCCL); out: host double[P] (synchronises). Either may be NULL. */
int vgan_hc_finalize(vgan_hc_ctx *c, double *d_out, double *out);
int vgan_hc_synchronize(vgan_hc_ctx *c);
/* Several GPUs in one process (the reference's OpenMP loop with its critical-section accumulate, HaploCart.cpp:408-421, one
 * context per GPU instead of one thread per core): every context finalizes, then ONE reduce of the P doubles onto the
 * first context -- ncclReduce over xGMI when the contexts sit on distinct devices (the communicator is created once per
 * device set, cached for the life of the process and reused by every later reduce over the same devices; RCCL is loaded at
 * run time, the library does not link it), through the host when contexts share a device, when RCCL cannot be loaded or
 * initialised, or when VGAN_HC_REDUCE=host is set in the environment (for a single reduce of 41 KB per context the cheaper
 * way: vgan_hc_reduce_info / _last report what the collective's set-up and the reduce cost).  out: host double[P] = the sum of
 * what all the contexts accumulated.  *used_rccl (or NULL): 1 if the collective ran. */
int vgan_hc_reduce(vgan_hc_ctx **ctxs, int n, double *out, int *used_rccl);
/* wall time of the last communicator set-up (ncclCommInitAll) and how many there were in this process; either may be NULL */
int vgan_hc_reduce_info(double *last_setup_ms, int *n_setups);
/* wall time of the last vgan_hc_reduce (a communicator set-up made inside it included) and which way it went; either may be NULL */
int vgan_hc_reduce_last(double *reduce_ms, int *was_rccl);
/* (ABI 5) Which way the last vgan_hc_reduce went, in words: returns 1 when it was an RCCL reduce (buf = ""), 0 when the vectors were
 * summed on the host, with the reason in buf (contexts sharing a device, VGAN_HC_REDUCE=host, librccl missing, ncclCommInitAll's
 * error string).  The two ways add the contexts' vectors in different orders: the sums may differ in their last bits. */
int vgan_hc_reduce_why(char *buf, int64_t cap);
void vgan_hc_destroy(vgan_hc_ctx *c);

/* Per-kernel device timing with HIP events on the context's stream (bench.py's roofline figure).
 * Slots: 0 segment kernel, 1 per-segment mask sweep, 2 per-node mask sweep, 3 finish, 4 the layout pass when vgan_hc_accumulate
 * has to run it (a batch without .packed). */
enum { VGAN_HC_K_SEGMENT = 0, VGAN_HC_K_SWEEP_SEG = 1, VGAN_HC_K_SWEEP_NODE = 2, VGAN_HC_K_FINISH = 3, VGAN_HC_K_PACK = 4,
       VGAN_HC_K_COUNT = 5 };
/* enable: 0 off, 1 events around every kernel, 2 around the segment kernel alone (a pair of events is ~8 us of the stream's time: a
 * caller timing whole steps at the same time -- bench.py -- asks for the one kernel its roofline is about).  Also clears the counters. */
int vgan_hc_profile_enable(vgan_hc_ctx *c, int enable);
/* synchronises the stream; ms[i] = summed device time of kernel i, launches[i] = number of launches timed */
int vgan_hc_profile_read(vgan_hc_ctx *c, double ms[5], uint64_t launches[5]);

/* get_posterior (get_posterior.cpp:87-127) on the device: log-sum-exp over the P paths and over the descendants of
 * each ancestor of `predicted`, gathered as get_posterior_of_clade does (:51-76): one child set per recursion level,
 * so on a children.txt that is not a tree a path reached at two depths is summed twice.  Where the reference is
 * undefined: an ancestor without path-name descendants sums nothing = 0 (confidence exp(0 - total)), a name absent
 * from children.txt has no children, a cyclic children.txt is an error.  clades: '\n'-joined names into clade_buf;
 * conf[i] beside it.  Returns the number of records (>= 1) or a negative error. */
int vgan_hc_posterior(vgan_hc_ctx *c, const double *final_vec /* host [P] */, const char *predicted,
                      char *clade_buf, int64_t clade_cap, double *conf, int32_t conf_cap);
/* argmax with the reference's first-maximum tie rule (std::max_element, HaploCart.cpp:423) -- with a tie taken as the arithmetic
 * means it: paths the reads do not tell apart (identical over every node a read touches) have EQUAL sums in exact arithmetic and sums
 * that differ in their last bits in floating point, by the order of the additions -- which under the reference's OpenMP loop, and
 * under this library's atomics, is not the same from run to run.  The first path within 1e-12 (relative) of the maximum is returned:
 * what std::max_element gives on the exact sums, the same name on every run.  (One column's term moves a sum by 1e-10 and more of
 * its size: paths a read does tell apart are not within the tolerance.) */
int vgan_hc_argmax(const double *final_vec, uint32_t n_paths);

/* ------------------------------------------------------------------------------------------------
 * euka: per-read two-model likelihood with ancient-DNA damage (readGAM3's per-alignment lambda,
 * readGAM_Euka.h:67-577; Damage::initDeamProbabilities, damage.cpp:41-323; Baseshift::baseshift_calc,
 * baseshift.cpp:57-88).  Replaces the call `readGAM3(...)` at Euka.cpp:534-537.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_euka_db vgan_euka_db; /* opaque: clade table + bins */

typedef struct vgan_euka_db_view {
    uint32_t n_clades;
    const int32_t *clade_id;      /* *.clade column 0 (load.cpp:118-152) */
    const double *clade_dist;     /* column 2: pairwise average distance */
    const int32_t *clade_npaths, *clade_snode, *clade_enode;
    const char *clade_names;      /* '\n' joined */
    const uint32_t *bin_off;      /* [n_clades+1] bins of clade c: *.bins line c (load.cpp:71-99) */
    const int32_t *bin_lo, *bin_hi; /* node id range, stoi of "1836.0"-style tokens */
    const double *bin_entropy;
} vgan_euka_db_view;

int vgan_euka_db_load(const char *clade_path, const char *bins_path, vgan_euka_db **out); /* plain or .gz */
int vgan_euka_db_from_arrays(const vgan_euka_db_view *v, vgan_euka_db **out);
int vgan_euka_db_view_get(const vgan_euka_db *d, vgan_euka_db_view *out);
void vgan_euka_db_free(vgan_euka_db *d);

typedef struct vgan_damage vgan_damage; /* opaque: 5' / 3' substitution matrices per position */
typedef struct vgan_damage_view {
    uint32_t n5, n3;      /* rows kept: positions >= n-1 repeat the last row (damage.cpp:91-93,134-136) */
    const double *sub5p;  /* [n5][4][4] row-stochastic, original base x damaged base (damage.cpp:66-88) */
    const double *sub3p;  /* [n3][4][4] */
} vgan_damage_view;
/* 12-column .prof text (header A>C ... T>G; miscfunc.h:84-136); NULL or "" = no damage (damage.cpp:47-55) */
int vgan_damage_from_text(const char *prof5_text, const char *prof3_text, vgan_damage **out);
int vgan_damage_load(const char *prof5_path, const char *prof3_path, vgan_damage **out);
int vgan_damage_view_get(const vgan_damage *d, vgan_damage_view *out);
void vgan_damage_free(vgan_damage *d);

/* SoA batch of the front half (reconstruct_graph_sequence + what the lambda reads off the Alignment). */
typedef struct vgan_euka_batch {
    uint32_t n_reads;
    uint64_t n_cols, n_qual, n_maps;
    const uint32_t *read_col_off;  /* [n_reads+1] region of graph_seq / read_seq (zero padded to the longer) */
    const uint32_t *read_qual_off; /* [n_reads+1] */
    const uint32_t *read_map_off;  /* [n_reads+1] */
    const uint16_t *read_gseq_len; /* [n_reads] |graph_seq| */
    const uint16_t *read_rseq_len; /* [n_reads] |read_seq| */
    const uint16_t *read_seq_len;  /* [n_reads] a.sequence().size() = Lseq, within 15..1000 */
    const int32_t *read_mapq;      /* [n_reads] */
    const uint8_t *read_rev;       /* [n_reads] first mapping is_reverse */
    const uint32_t *read_src;      /* [n_reads] index of the read in the alignment set */
    const uint32_t *map_node;      /* [n_maps] node id of every mapping */
    const uint8_t *graph_seq, *read_seq, *qual;
    int32_t on_device;
    uint32_t reserved;
} vgan_euka_batch;

typedef struct vgan_euka_host_batch vgan_euka_host_batch;
typedef struct vgan_euka_flatten_stats {
    int64_t n_in, n_out, n_unmapped /* identity == 0, readGAM_Euka.h:72 */, n_bad /* reference reads out of bounds */;
} vgan_euka_flatten_stats;
int vgan_euka_flatten(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, int n_threads,
                      vgan_euka_host_batch **out, vgan_euka_flatten_stats *stats);
int vgan_euka_host_batch_get(const vgan_euka_host_batch *b, vgan_euka_batch *out);
void vgan_euka_host_batch_free(vgan_euka_host_batch *b);

typedef struct vgan_euka_params {
    uint32_t min_mapq;       /* MINIMUMMQ, --minMQ (Euka.cpp:152-190: 29) */
    int32_t length_to_prof;  /* -l (5); at most 32 */
} vgan_euka_params;

/* per-read results (arrays of n_reads, host memory for host batches, device memory for device batches) */
typedef struct vgan_euka_read_out {
    int32_t *clade;     /* c_n (line index in *.clade), -1 = the reference would index out of bounds on this read */
    double *in_lik;     /* in_clade_lik */
    double *out_lik;    /* not_in_clade_lik */
    double *like;       /* value pushed to Clade::clade_like (readGAM_Euka.h:491) */
    double *not_like;   /* ... Clade::clade_not_like (:492) */
    uint8_t *pass;      /* in - out > 1 && mapq > MINIMUMMQ (:504-510) */
} vgan_euka_read_out;

typedef struct vgan_euka_ctx vgan_euka_ctx;
int vgan_euka_create(const vgan_euka_db_view *db, const vgan_damage_view *dmg, const vgan_euka_params *prm, int device,
                     vgan_euka_ctx **out);
int vgan_euka_set_stream(vgan_euka_ctx *c, void *hip_stream);
int vgan_euka_reset(vgan_euka_ctx *c);
int vgan_euka_accumulate(vgan_euka_ctx *c, const vgan_euka_batch *b, const vgan_euka_read_out *out);
/* per-clade results of everything accumulated: Clade::count [n_clades], baseshift_clade_array
 * [n_clades][2*length_to_prof][16], bin coverage get<3>(chunks[c][j]) [n_bins]; host arrays; synchronises */
int vgan_euka_synchronize(vgan_euka_ctx *c); /* (ABI 6) waits for the context's stream: a device batch's kernel is asynchronous */
int vgan_euka_finalize(vgan_euka_ctx *c, int32_t *clade_count, uint32_t *baseshift, double *bin_cov, int64_t *n_bad);
int vgan_euka_kernel_ms(vgan_euka_ctx *c, double *ms, uint64_t *launches); /* HIP-event time of the read kernel */
void vgan_euka_destroy(vgan_euka_ctx *c);

/* sum over the reads of clade c of log(clade_like[k]) and their number, for everything accumulated (valid after
 * vgan_euka_finalize): all that MCMC::get_proposal_likelihood (MCMC.cpp:1175-1215) reads of clade_like / clade_not_like,
 * because its `(1/334)` is the integer 0 and log(frac * like) = log(frac) + log(like).  A read with like == 0 (mapq 0, or
 * exp underflow) makes the clade's sum -inf, as it does the reference's. */
int vgan_euka_like_sums(vgan_euka_ctx *c, int64_t *n_like, double *sum_log_like);
/* Several contexts (one per GPU, the reads dealt between them): vgan_euka_finalize + vgan_euka_like_sums of every context,
 * summed -- the integer tables exactly.  Any output may be NULL.  (The per-read outputs of vgan_euka_accumulate are the
 * caller's to put back in input order: readGAM3 hands them to the abundance chain per read, MCMC.cpp:1192-1193.) */
int vgan_euka_reduce(vgan_euka_ctx **ctxs, int n, int32_t *clade_count, uint32_t *baseshift, double *bin_cov, int64_t *n_like,
                     double *sum_log_like, int64_t *n_bad);

/* ------------------------------------------------------------------------------------------------
 * euka downstream of the per-read pass (host only): the detected-clade list at the end of readGAM3
 * (readGAM_Euka.h:582-630), Euka::compute_init_vec (compute_init_vec.cpp:9-84), the abundance MCMC
 * (MCMC::generate_proposal / get_proposal_likelihood / run, MCMC.cpp:1095-1366) and the output files of
 * Euka::run (Euka.cpp:540-1160).
 *
 * Randomness: the reference seeds a fresh std::mt19937 from std::random_device for every proposal and once for the
 * acceptance draws, so its output is not reproducible.  seed = 0 does the same; any other seed replaces the successive
 * random_device calls by the high 32 bits of a splitmix64 stream started at seed (run() first, then one per proposal).
 * ------------------------------------------------------------------------------------------------ */
/* (ABI 6) euka's front half on the device (csrc/euka_flatten_kernels.hip; reference: src/readGAM_Euka.h:67-216 through
 * csrc/host/euka_host.cpp): the arrays a vgan_gamdev parse left in HBM -> a vgan_euka_batch in HBM (on_device = 1, valid until the next run
 * on the object; read_src is a DEVICE array here: vgan_euka_devflat_host_arrays has its host copy and read_seq_len's), for the reads whose
 * edits are all matches or substitutions on known nodes and whose lengths euka's tables hold -- array for array vgan_euka_flatten's batch
 * of those reads.  Every other read is left to the host: host_mask[r] = 1 (caller array, one byte per read of the parse).  base: index of
 * the parse's first read in the whole input (read_src = base + index).  Runs on the context's stream; synchronises it. */
typedef struct vgan_euka_devflat vgan_euka_devflat;
struct vgan_gamdev;
int vgan_euka_devflat_create(vgan_euka_ctx *c, const vgan_graph *graph, vgan_euka_devflat **out);
int vgan_euka_devflat_run_gamdev(vgan_euka_devflat *f, const struct vgan_gamdev *gd, uint32_t base, vgan_euka_batch *out, uint8_t *host_mask,
                                 vgan_euka_flatten_stats *stats);
int vgan_euka_devflat_host_arrays(const vgan_euka_devflat *f, const uint32_t **read_src, const uint16_t **read_seq_len);
void vgan_euka_devflat_free(vgan_euka_devflat *f);
/* (test aid) a device batch's arrays copied into the caller's arrays (the pointers of *host; any may be NULL) */
int vgan_euka_batch_download(const vgan_euka_batch *dev, const vgan_euka_batch *host);
/* (ABI 6) vgan euka over a BGZF GAM's bytes through the device front end's pipeline (vgan_gampipe_*: the file in pieces, piece i to lane
 * i mod n; reference: readGAM_Euka.h:581 feeding the lambda of :67-577): start / attach / finish as vgan_hc_gam_*.  The contexts hold the
 * per-clade tables afterwards (vgan_euka_reduce); the per-read lists the abundance chain and the report read come back here, in the order
 * of the file: read_index = the read's index among the file's mapped reads (identity != 0, readGAM_Euka.h:72), for the reads that were
 * processed (n_reads of them; n_bad more were skipped: the reference would index out of bounds on them).  The arrays belong to the run
 * until vgan_euka_gam_free. */
typedef struct vgan_euka_gamrun vgan_euka_gamrun;
typedef struct vgan_euka_gam_result {
    int64_t n_messages; /* alignments in the file */
    int64_t n_mapped;   /* identity != 0 */
    int64_t n_bad, n_reads;
    const uint32_t *read_index;
    const int32_t *read_clade;
    const uint8_t *read_pass;
    const uint16_t *read_seq_len;
} vgan_euka_gam_result;
int vgan_euka_gam_start(const int *devices, int n_lanes, const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, vgan_euka_gamrun **out);
int vgan_euka_gam_attach(vgan_euka_gamrun *r, vgan_euka_ctx *const *ctxs, int n_ctx, const vgan_graph *graph);
int vgan_euka_gam_finish(vgan_euka_gamrun *r, vgan_euka_gam_result *res, vgan_gampipe_stats *pstats);
void vgan_euka_gam_free(vgan_euka_gamrun *r);

typedef struct vgan_euka_detect_params {
    uint32_t min_bins;        /* MINNUMOFBINS  --minBins (6), Euka.cpp:183 */
    uint32_t min_reads;       /* MINNUMOFREADS --minFrag (10) */
    int32_t max_zero_bins;    /* MAXIMUMOFBINS --maxBins (0) */
    double entropy_threshold; /* ENTROPY_SCORE_THRESHOLD --entropy (1.17) */
} vgan_euka_detect_params;
/* ids[] (capacity n_clades) receives Clade::id of every detected clade in *.clade order.  A bin counts as empty when its
 * coverage truncates to 0 (the reference collects the coverages in a vector<int>); the last bin of a clade is ignored. */
int vgan_euka_detect(const vgan_euka_db_view *db, const int32_t *clade_count, const double *bin_cov,
                     const vgan_euka_detect_params *p, int32_t *ids, int32_t *n_ids);
/* est[5*n] = per clade {median of the recorded proposals, 15 %, 85 %, 5 %, 95 % quantile} (MCMC.cpp:1318-1360).
 * init[n] = starting abundances, n_like / sum_log_like[n] = vgan_euka_like_sums rows of the n detected clades.
 * iter must exceed burnin + 1 (the reference indexes an empty sample otherwise). */
int vgan_euka_abundance_mcmc(int32_t n, const double *init, const int64_t *n_like, const double *sum_log_like, int32_t iter,
                             int32_t burnin, uint64_t seed, double *est);

typedef struct vgan_euka_report_cfg {
    vgan_euka_detect_params detect;
    int32_t length_to_prof;  /* -l, the baseshift array's lengthToProf */
    int32_t run_mcmc;        /* 0 = --no-mcmc (also skipped with fewer than two detected clades, Euka.cpp:586) */
    int32_t iter, burnin;    /* --iter (10000), --burnin (100) */
    uint64_t seed;
    int32_t out_frag;        /* --outFrag: <prefix>_FragNames.tsv */
    uint32_t reserved;
    const char *out_group;   /* --outGroup, NULL or "" = none */
    const char *out_dir;     /* --out_dir: created when missing (Euka.cpp:738-747), NULL = none */
} vgan_euka_report_cfg;

typedef struct vgan_euka_results { /* what readGAM3 leaves behind (Euka.cpp:534-537) */
    const vgan_euka_db_view *db;
    const int32_t *clade_count;   /* vgan_euka_finalize */
    const uint32_t *baseshift;
    const double *bin_cov;
    const int64_t *n_like;        /* vgan_euka_like_sums */
    const double *sum_log_like;
    int64_t n_reads;              /* per processed read, in input order: feeds Clade::inSize / nameStorage */
    const int32_t *read_clade;    /* vgan_euka_read_out.clade */
    const uint8_t *read_pass;
    const uint16_t *read_seq_len; /* vgan_euka_batch.read_seq_len */
    const int64_t *name_off;      /* [n_reads+1] into names; NULL unless out_frag */
    const char *names;
} vgan_euka_results;
/* Writes <prefix>_{abundance,detected,coverage,inSize}.tsv, <prefix>_<clade>.prof per detected clade (and the out
 * group), <prefix>_{5p,3p}.prof and, with out_frag, <prefix>_FragNames.tsv, byte for byte as Euka::run formats them.
 * detected[] (capacity n_clades + 1) / estimates[5 per detected clade] may be NULL. */
int vgan_euka_report(const vgan_euka_results *r, const vgan_euka_report_cfg *cfg, const char *prefix, int32_t *detected,
                     int32_t *n_detected, double *estimates);

/* ------------------------------------------------------------------------------------------------
 * soibean: per-read x per-path likelihood (analyse_GAM, getLCAfromGAM.h:31-732; replaces the call at
 * soibean.cpp:558) and the per-MCMC-iteration likelihood refresh (MCMC.cpp:738-993 inside
 * MCMC::run_tree_proportion, MCMC.h:87; computeBaseLogLike MCMC.h:111-296).
 *
 * The reference keeps, per read and path, a list of per-base records (detailMap: reads x paths x bases x 16 B) and
 * re-walks it every iteration.  Here analyse_GAM's result is factorised once into, per (read, path):
 *   pm   = sum of the per-base log-likelihoods (pathMap)
 *   cnt  = 5x5 counts of (reference base, read base) over the path-supported bases (the only bases that get the HKY
 *          term, which depends on nothing but that pair and the branch time)
 * so one refresh is  LL(read, path, t) = pm + sum_j cnt[j] * hky_t[j]  -- a streaming reduction over HBM-resident
 * tables laid out [path][j][read] so that the 2k paths an iteration touches are read fully coalesced.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_sb_batch {
    uint32_t n_reads, n_segments;
    uint64_t n_cols, n_qual;
    const uint32_t *read_seg_off;  /* [n_reads+1] edit-level segments (mppg_sizes entries, getLCAfromGAM.h:139) */
    const uint32_t *read_col_off;  /* [n_reads+1] */
    const uint32_t *read_qual_off; /* [n_reads+1] */
    const uint16_t *read_gseq_len; /* [n_reads] |graph_seq| = Lseq of the damage table (getLCAfromGAM.h:109), 15..1000 */
    const uint16_t *read_rseq_len; /* [n_reads] |read_seq| */
    const uint8_t *read_rev;       /* [n_reads] */
    const uint32_t *read_src;      /* [n_reads] index in the alignment set */
    const uint32_t *seg_node;      /* [n_segments] node id of mapping i, 0 = "No_support" (:156-160) */
    const uint16_t *seg_col;       /* [n_segments] first column of the slice (baseIX, or startIndex on the reverse strand) */
    const uint16_t *seg_len;       /* [n_segments] |nodeSeq| */
    const uint16_t *seg_base_ix;   /* [n_segments] baseIX: damage position and start of the penalty pattern */
    const uint8_t *graph_seq, *read_seq, *qual;
    int32_t on_device;
    uint32_t reserved;
} vgan_sb_batch;

typedef struct vgan_sb_host_batch vgan_sb_host_batch;
typedef struct vgan_sb_flatten_stats {
    int64_t n_in, n_out, n_unmapped, n_bad;
} vgan_sb_flatten_stats;
int vgan_sb_flatten(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, int n_threads,
                    vgan_sb_host_batch **out, vgan_sb_flatten_stats *stats);
int vgan_sb_host_batch_get(const vgan_sb_host_batch *b, vgan_sb_batch *out);
void vgan_sb_host_batch_free(vgan_sb_host_batch *b);

typedef struct vgan_sb_params {
    int32_t penalty; /* PENALTY, -P (soibean.cpp: 7) */
    int32_t reserved;
} vgan_sb_params;

typedef struct vgan_sb_source { /* one source of an MCMC state (MCMC.cpp:885-980) */
    int32_t child, parent;  /* path indices: pathNames[y], parentpathNames[y] */
    double dist;            /* positions_tree[y].pos->dist, the child's branch length (0 -> 1e-5) */
    double pos;             /* positions_tree[y].pos_branch */
    double theta;           /* proportions[y] (ignored when k = 1) */
} vgan_sb_source;

typedef struct vgan_sb_ctx vgan_sb_ctx;
/* graph view: mask = nodepaths (paths through each node), path_names for the 101-character rule; at most 256 paths */
int vgan_sb_create(const vgan_graph_view *graph, const vgan_damage_view *dmg, const vgan_sb_params *prm, int device,
                   vgan_sb_ctx **out);
int vgan_sb_set_stream(vgan_sb_ctx *c, void *hip_stream);
/* analyse_GAM over the batch: (re)builds the device-resident factorised tables.  n_bad: reads excluded on the device */
int vgan_sb_precompute(vgan_sb_ctx *c, const vgan_sb_batch *b, int64_t *n_bad);
/* (ABI 7) soibean's front half on the device (csrc/sb_flatten_kernels.hip; reference: src/getLCAfromGAM.h:92-186,537-544): the arrays of
 * the last vgan_gamdev_parse -> rows of a vgan_sb_batch in HBM, APPENDED to the batch the object holds (analyse_GAM runs once over a
 * context's reads), in input order, for the reads whose edits are all matches or substitutions on known nodes and whose slices cannot
 * leave the strings (the conditions are beside sb_df_classify_kernel); host_mask[r] = 1 for every other read of the parse: the host
 * flattens those (vgan_sb_flatten) and appends its batch (vgan_sb_devflat_append_host; src_map: read_src[i] -> src_map[read_src[i]]).
 * Array for array vgan_sb_flatten's batch of the same reads.  base: the index of the parse's first read (read_src = base + r).
 * vgan_sb_devflat_batch: the batch so far (device pointers, on_device = 1; valid until the next append).  vgan_sb_devflat_expect: the
 * first append sizes the arrays for `scale` times what it brings.  vgan_sb_batch_download: a device batch's arrays into host arrays. */
typedef struct vgan_sb_devflat vgan_sb_devflat;
struct vgan_gamdev;
int vgan_sb_devflat_create(vgan_sb_ctx *c, const vgan_graph *graph, vgan_sb_devflat **out);
int vgan_sb_devflat_expect(vgan_sb_devflat *f, double scale);
int vgan_sb_devflat_append_gamdev(vgan_sb_devflat *f, const struct vgan_gamdev *gd, uint32_t base, uint8_t *host_mask, vgan_sb_flatten_stats *stats);
int vgan_sb_devflat_append_host(vgan_sb_devflat *f, const vgan_sb_batch *host_batch, const uint32_t *src_map);
int vgan_sb_devflat_batch(const vgan_sb_devflat *f, vgan_sb_batch *out);
void vgan_sb_devflat_free(vgan_sb_devflat *f);
int vgan_sb_batch_download(const vgan_sb_batch *dev, const vgan_sb_batch *host);
/* `vgan soibean` over the device front end's pipeline (csrc/sb_gam_run.hip; vgan_hc_gam_* / vgan_euka_gam_* are its siblings): start
 * begins upload, inflate, framing and the protobuf walk of the file's pieces at once (they need no context); attach gives the contexts
 * (one per lane) and the graph; finish waits for the pieces, appends the host's share and makes analyse_GAM's tables once per context
 * (vgan_sb_precompute over the context's whole batch) -- the contexts are then as after the host pipeline's vgan_sb_precompute calls,
 * with the reads dealt by pieces instead of contiguous shares (the chains' sums are integers: the same bits).  n_bad: reads of the
 * host's share that vgan_sb_flatten refuses; n_dev_bad: vgan_sb_precompute's.  vgan_sb_gam_batch: lane `lane`'s batch (device
 * pointers; read_src = the reads' places among the file's mapped reads) until vgan_sb_gam_free. */
typedef struct vgan_sb_gamrun vgan_sb_gamrun;
typedef struct vgan_sb_gam_result {
    int64_t n_messages, n_mapped, n_reads, n_bad, n_dev_bad;
    double ms_tables; /* appending the host's share + vgan_sb_precompute */
} vgan_sb_gam_result;
int vgan_sb_gam_start(const int *devices, int n_lanes, const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, vgan_sb_gamrun **out);
int vgan_sb_gam_attach(vgan_sb_gamrun *r, vgan_sb_ctx *const *ctxs, int n_ctx, const vgan_graph *graph);
int vgan_sb_gam_finish(vgan_sb_gamrun *r, vgan_sb_gam_result *res, vgan_gampipe_stats *pstats);
int vgan_sb_gam_batch(const vgan_sb_gamrun *r, int lane, vgan_sb_batch *out);
void vgan_sb_gam_free(vgan_sb_gamrun *r);
/* test aid: pm [n_paths][r1-r0], cnt [n_paths][25][r1-r0], ok [r1-r0] of the resident reads, host arrays */
int vgan_sb_read_tables(vgan_sb_ctx *c, uint32_t r0, uint32_t r1, double *pm, uint16_t *cnt, uint8_t *ok);
/* n_states likelihood refreshes in one launch (e.g. the independent chains, soibean.cpp:805-840); src has n_states*k
 * entries; freqs7 = {A, C, G, T, R, Y, M} (soibean.cpp:609-640); out: host double[n_states] (synchronises) and/or
 * d_out: device double[n_states] for an RCCL all-reduce.  guard[n_states] (host, optional) counts reads on which one
 * of the reference's runtime_error guards would fire. */
int vgan_sb_loglike(vgan_sb_ctx *c, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                    double *out, double *d_out, uint64_t *guard);
/* (ABI 4) Every sum over reads of this path (a refresh, the initial mixture) is taken in fixed point, in integers: units of
 * 2^-44, value = (hi * 2^32 + lo) * 2^-44 + nf, a term that is not finite or 2^18 and beyond in magnitude going into the plain
 * double nf.  Integer sums do not depend on the order of their terms, so a log-likelihood is the same bits however the reads
 * are dealt to lanes, workgroups, contexts or GPUs -- an MCMC accept / reject does not depend on the number of devices.  A
 * caller holding the reads in several contexts adds the contexts' sums (vgan_sb_sum_add) and converts once
 * (vgan_sb_sum_value); vgan_sb_group_* below does that for the chain driver. */
typedef struct vgan_sb_sum {
    int64_t hi;
    uint64_t lo;
    double nf;
} vgan_sb_sum;
double vgan_sb_sum_value(const vgan_sb_sum *s);
void vgan_sb_sum_add(vgan_sb_sum *acc, const vgan_sb_sum *x);
/* vgan_sb_loglike / vgan_sb_mixture_loglike returning the sums themselves (host, synchronises) */
int vgan_sb_loglike_sums(vgan_sb_ctx *c, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                         vgan_sb_sum *sums, uint64_t *guard);
/* analyse_GAM's per-read mostProbPath (getLCAfromGAM.h:563-579) over the resident reads: best[r] (host, n_reads of the
 * batch, may be NULL) = the path holding the read's highest pathMap value when exactly one path does, -1 on a tie or for a
 * read excluded on the device; sig_count[n_paths] = reads per uniquely best path, the "signature" frequencies of
 * soibean.cpp:655-668; n_reads_ok = gam->size().  Paths with the same support pattern over a read have bit-identical sums,
 * so the reference's ties are ties here. */
int vgan_sb_best_paths(vgan_sb_ctx *c, int32_t *best, int64_t *sig_count, int64_t *n_reads_ok);
/* the initial log-likelihood of soibean.cpp:737-756 over the resident reads: sum_r (+)_j (log_freq + pathMap_r[paths[j]])
 * with oplusInitnatl as (+); one source and log_freq = 0 gives the plain sum of :744-747 */
int vgan_sb_mixture_loglike(vgan_sb_ctx *c, uint32_t n, const int32_t *paths, double log_freq, double *out);
int vgan_sb_mixture_sums(vgan_sb_ctx *c, uint32_t n, const int32_t *paths, double log_freq, vgan_sb_sum *sum);
/* host: the initial sources of soibean.cpp:669-712 from the signature counts: paths with at least 1 % of the reads, by
 * descending count (equal counts: ascending path index; the reference's order among them is that of an unordered_map),
 * cut to cutk when cutk > 0; every path with a count when none reaches the threshold.  paths[] capacity n_paths */
int vgan_sb_signature_paths(const int64_t *sig_count, uint32_t n_paths, int64_t n_reads, int32_t cutk, int32_t *paths, int32_t *n);
int vgan_sb_kernel_ms(vgan_sb_ctx *c, double ms[2], uint64_t launches[2]); /* 0 precompute kernel, 1 refresh kernel */
void vgan_sb_destroy(vgan_sb_ctx *c);

/* ------------------------------------------------------------------------------------------------
 * soibean downstream of analyse_GAM (host control flow around vgan_sb_loglike): the taxon tree
 * (<dbprefix>.new.dnd, soibean.cpp:565-596), MCMC::run_tree_proportion (MCMC.cpp:522-1093) with MCMC::updatePosition
 * (:169-470) and MCMC::sample_normal (:487-520), MCMC::processMCMCiterations (:23-150) and the chain loop with its R-hat
 * diagnostics (soibean.cpp:738-944).
 *
 * Where the reference leaves the behaviour open this build defines it:
 *  - randomness: every std::random_device call is replaced by the next output of a splitmix64 stream started at `seed`
 *    (0 = the hardware source, as the reference); libc rand() -- never seeded there -- and the function-local static engine
 *    of sample_normal by one std::mt19937 each per chain, seeded from that stream when the chain starts (before the chain's
 *    own engine).  Chains therefore do not share random state: vgan_sb_estimate advances the chains of a source count
 *    together and evaluates their proposed states in one likelihood call (one launch) per iteration;
 *  - tree nodes are numbered in pre-order of the Newick text (spidir, the reference's tree library, is not in its tree);
 *  - getPatristicDistances indexes a vector of #leaves entries by node index: only indices below #leaves are compared;
 *  - diagnostics rows come in branch-name order (an unordered_map there); a branch a chain did not end on takes the
 *    reference's own defaults {1, 1, 1, 1} for that chain.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_tree vgan_tree; /* opaque */
typedef struct vgan_tree_view {
    uint32_t n_nodes, n_leaves;
    int32_t root;
    const int32_t *parent;    /* [n_nodes], -1 at the root */
    const double *dist;       /* [n_nodes] branch length above the node (0 when absent) */
    const int32_t *child_off; /* [n_nodes+1] into children */
    const int32_t *children;
    const char *names;        /* '\n' joined node labels (spidir longname) */
} vgan_tree_view;
int vgan_tree_parse(const char *newick, vgan_tree **out);
int vgan_tree_load(const char *path, vgan_tree **out); /* plain or .gz */
int vgan_tree_view_get(const vgan_tree *t, vgan_tree_view *out);
void vgan_tree_free(vgan_tree *t);

/* what the chain needs of the likelihood: one refresh (MCMC.cpp:738-993) and the initial mixture (soibean.cpp:737-756).
 * Return VGAN_OK or a negative code; guard counts reads on which the reference would throw. */
typedef struct vgan_sb_engine {
    void *user;
    int (*refresh)(void *user, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7, double *loglike, uint64_t *guard);
    int (*mixture)(void *user, uint32_t n, const int32_t *paths, double log_freq, double *loglike);
    /* optional (may be NULL): n_states states of k sources each in one call; src[n_states * k], loglike / guard[n_states].
     * vgan_sb_estimate advances the chains of one source count together and asks for all their states at once. */
    int (*refresh_many)(void *user, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                        double *loglike, uint64_t *guard);
} vgan_sb_engine;
int vgan_sb_engine_gpu(vgan_sb_ctx *c, vgan_sb_engine *out); /* vgan_sb_loglike / vgan_sb_mixture_loglike of the context */
/* (ABI 4) The reads of one job dealt to several contexts -- one per GPU; MCMC.cpp:739 is an OpenMP loop over the reads with
 * `reduction(+:logLike)`, here the loop runs over devices -- as ONE engine: a refresh is launched on every context before any
 * is waited for, the contexts' sums (vgan_sb_sum: integers) are added on the host and converted once.  The chain files are
 * those of one context holding all the reads, byte for byte.  The contexts stay the caller's (the group only refers to them). */
typedef struct vgan_sb_group vgan_sb_group;
int vgan_sb_group_create(vgan_sb_ctx **ctxs, int n, vgan_sb_group **out);
void vgan_sb_group_free(vgan_sb_group *g);
int vgan_sb_engine_group(vgan_sb_group *g, vgan_sb_engine *out);
/* vgan_sb_best_paths' signature counts and usable-read count summed over the group's contexts */
int vgan_sb_group_best_paths(vgan_sb_group *g, int64_t *sig_count, int64_t *n_reads_ok);
/* the engine's refresh is one fused kernel plus a fold into pinned host memory; on != 0 brackets it with HIP events so that
 * vgan_sb_kernel_ms (slot 1) reports it as well -- off by default: the chain is launch bound and two event records cost */
int vgan_sb_time_engine(vgan_sb_ctx *c, int on);
/* The engine's refresh as a RESIDENT kernel (csrc/sb_kernels.hip: sb_refresh_resident_kernel), off unless asked for (on = 1 here, or
 * VGAN_SB_RESIDENT=1 in the environment when the context is made).  MCMC.cpp:738-993 wants one likelihood per iteration and cannot go on
 * without it; with this on the first refresh starts a kernel that stays: the host writes a refresh's sources into a pinned mailbox, the
 * kernel answers into pinned memory -- no launch per iteration.  Measured (soibean): 21 us instead of 24 per iteration up to
 * ~30 k reads, no gain from there to ~250 k, slower beyond (the kernel keeps half the device free for whatever else the process runs).
 * One context per device has the kernel at a time (another context's refreshes are launched); the kernel leaves when any other call on
 * the context needs the device's tables (the next refresh starts it again) and by itself after 5 ms without a refresh; a refresh timed
 * with HIP events (vgan_sb_time_engine) is a launched one.  Results are the launched refresh's, bit for bit.  on < 0: returns the setting.
 * vgan_sb_resident_launches: how often the kernel was started (test aid).  vgan_sb_kernel_ms slot 1 then counts the kernel's own clock
 * from seeing a refresh to publishing it (read when the kernel leaves: the call makes it leave). */
int vgan_sb_resident(vgan_sb_ctx *c, int on);
int vgan_sb_resident_launches(const vgan_sb_ctx *c, uint64_t *launches);

typedef struct vgan_sb_estimate_cfg {
    uint32_t max_iter;  /* --iter (500000) */
    uint32_t burn;      /* --burnin (75000), below max_iter */
    uint32_t chains;    /* --chains (4) */
    uint32_t n_paths;   /* paths of the graph = tree nodes: sizes the proposal (MCMC.cpp:541-546) and the random starts */
    uint64_t seed;
    double con;         /* shortest non-zero branch below 1, else 0.01 (soibean.cpp:598-602) */
    double freqs7[7];   /* A, C, G, T, R, Y, M (soibean.cpp:609-640) */
    int32_t run_mcmc;   /* 0 = --no-mcmc: initial log-likelihoods only */
    int32_t quiet;
} vgan_sb_estimate_cfg;
/* soibean.cpp:738-944 for the starting nodes sig_nodes[0..n_sig): for k = 1..n_sig sources the initial log-likelihood, then
 * `chains` chains (chain 0 from sig_nodes[0..k), the others from random nodes), each writing <prefix>Result<k><chain>.mcmc and
 * <prefix>Trace<k><chain>.detail.mcmc (gzip), appending to <prefix>ProportionEstimates<k>.txt / <prefix>BranchEstimate<k>.txt,
 * and <prefix>Diagnostics<k>0.txt.  node_path[n_nodes] = graph path index of every tree node (by name), -1 = none. */
int vgan_sb_estimate(const vgan_sb_engine *engine, const vgan_tree *tree, const int32_t *node_path, const int32_t *sig_nodes,
                     uint32_t n_sig, const vgan_sb_estimate_cfg *cfg, const char *out_prefix);

typedef struct vgan_synth_euka_cfg {
    uint64_t seed;
    uint32_t n_clades;        /* 335 */
    uint32_t nodes_per_clade; /* contiguous node-id range per clade */
    uint64_t n_reads;
    uint32_t read_len_mean;   /* 75, clipped 30..150 */
    uint32_t reserved;
    uint64_t read_seed;       /* reads only (0 = seed): ranks share the graph and draw different reads */
} vgan_synth_euka_cfg;
/* clade graph (nodes <= 5 bp), its clade/bin tables, and aDNA-like reads with the given damage applied */
int vgan_synth_euka(const vgan_synth_euka_cfg *cfg, const vgan_damage *dmg, vgan_graph **g, vgan_euka_db **db,
                    vgan_alnset **reads);

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs of the published shape: hcfiles-like graph + reads.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vgan_synth_graph_cfg {
    uint64_t seed;
    uint32_t genome_len;  /* 16569 */
    uint32_t n_nodes;     /* 11821 */
    uint32_t n_paths;     /* 5179 */
} vgan_synth_graph_cfg;

typedef struct vgan_synth_reads_cfg {
    uint64_t seed;
    uint64_t n_reads;
    uint32_t read_len;       /* 150 */
    double indel_rate;       /* 0.005 reads with a 1-3 bp indel */
    double softclip_rate;    /* 0.01 reads with a 5-20 bp softclip */
    double low_mapq_rate;    /* 0.1 reads with mapq U{0..59}, else 60 */
    int32_t errors;          /* 1: substitutions with prob 10^(-Q/10) */
    uint64_t first_read;     /* read i of the set is read first_read + i of the seed's stream: contiguous shards of one
                              * workload (bench.py's ranks, the sharding tests) are the same reads whatever the shard count */
} vgan_synth_reads_cfg;

int vgan_synth_hc_graph(const vgan_synth_graph_cfg *cfg, vgan_graph **out);
int vgan_synth_hc_reads(const vgan_graph *g, const vgan_synth_reads_cfg *cfg, vgan_alnset **out);

#ifdef __cplusplus
}
#endif
#endif
