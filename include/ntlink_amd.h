/*
 * ntlink_amd.h -- C ABI of the MI355X implementation of the ntLink `pair` hot path.
 *
 * The reference (bcgsc/ntLink v1.3.11) has no FFI for this path: its boundary is two
 * executables joined by a pipe (ntLink:221-225).  Each group of entry points below replaces
 * one of them; INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 *   ntl_batch_*   sequence hand-over                <- btllib SeqReader inside `indexlr`
 *                                                      (ntLink:199,223; id/seq semantics of
 *                                                      bin/read_fasta.py:6-46)
 *   ntl_sketch_*  (k,w) minimizer sketch            <- `indexlr --long --pos --strand [--len]
 *                                                      -k K -w W` (ntLink:199,223) and
 *                                                      btllib.Indexlr(path,k,w,LONG_MODE,t)
 *                                                      (bin/ntlink_patch_gaps.py:417-441)
 *   ntl_overlap_filter  valid-region + per-sequence   <- read_minimizer_line (bin/ntlink_overlap_sequences.py:170-190)
 *                 duplicate filter of the k15 w5 sketch
 *   ntl_index_*   contig minimizer index            <- NtLink.read_minimizers
 *                                                      (bin/ntlink_pair.py:189-211)
 *   ntl_map_*     per-read lookup, hit filtering,   <- NtLink.find_scaffold_pairs body
 *                 accepted contigs, PAF blocks         (bin/ntlink_pair.py:352-408),
 *                                                      get_accepted_anchor_contigs
 *                                                      (bin/ntlink_utils.py:200-294),
 *                                                      print_paf (bin/ntlink_paf_output.py:103-135)
 *   ntl_fastx_*   FASTA/FASTQ(.gz) records          <- `gzip -cd -f FILES |` + SeqReader
 *                                                      (ntLink:113-117,222-223)
 *   ntl_tsv_*     indexlr TSV -> arrays             <- the split()s of bin/ntlink_pair.py:197-207,355-378
 *   ntl_write_*   indexlr TSV / verbose / PAF text  <- indexlr stdout (ntLink:199,223),
 *                                                      bin/ntlink_pair.py:308-313,382-388,
 *                                                      bin/ntlink_paf_output.py:131-135
 *   ntl_tally_*   contig-pair tally                 <- bin/ntlink_pair.py:416-435,315-334,157-239
 *   ntl_liftover  verbose mappings through an AGP   <- bin/ntlink_liftover_mappings.py:61-143 (ntLink_rounds:124-125)
 *
 * Conventions: every call returns 0 on success or a negative NTL_E* code; the message is
 * available from ntl_last_error().  All pointers in signatures are HOST pointers unless the
 * name says otherwise; buffers are caller-owned; counts are obtained first, then filled.
 * Handles are opaque; one context per device; calls on one context are not thread-safe, except
 * ntl_host_alloc / ntl_host_free (a reader thread may take staging buffers while another drives the device)
 * and the host-only groups (ntl_fastx_*, ntl_tsv_*, ntl_write_*, ntl_tally_*: one object per thread).
 * Strands are encoded 1 = '+', 0 = '-'.  Results are in input order (reads, then minimizers in
 * position order), exactly as the reference emits them.
 *
 * Environment (tuning and tests; none is needed): NTL_IO_THREADS (parser threads, default
 * min(cores, 32), or one and a half per core of the process's cgroup CPU quota when that is less), NTL_IO_MIN_CHUNK (bytes per parser thread below which fewer threads are used),
 * NTL_IO_NO_MMAP=1 (stream every input through zlib on one thread), NTL_IO_PREAD=1 (plain files through staged preads
 * instead of a mapping), NTL_IO_NO_LIBDEFLATE=1, NTL_POOL_MAX_BYTES (bound of the cache of device blocks, default half the device memory),
 * NTL_IO_GZ_WHOLE_MAX (compressed bytes up to which a gzip file is inflated in one go, default
 * 1/40 of the physical memory within 1..16 GiB), NTL_IO_TRACE=1 (reader diagnostics on stderr), NTL_SKETCH_C / NTL_SKETCH_NT (k-mers per
 * lane, lanes per strip of the sketch kernel), NTL_SKETCH_FAST=0 (exact 64-bit window pass only), NTL_SKETCH_FORCE_REDO=1
 * (every strip takes the 32-bit passes and the exact pass), NTL_SKETCH_THRESH (0: the block-minima window pass for every
 * window instead of the threshold pass for 71 <= w <= 255; x: x candidates per window instead of 10), NTL_SKETCH_THRESH_DIRECT=1
 * (the threshold pass without staged keys for the large windows too), NTL_SKETCH_LANES=1 and NTL_EMIT_U=2 (kernel variants kept
 * for the record), NTL_PIPELINE=0 (one stream per context; default: a second stream for
 * the window stage), NTL_PIPELINE_PRIO (0 no stream priorities, 1 = default: MAIN above the window stream, 2 the reverse),
 * NTL_SKETCH_CAP_GUESS (records a sketch's arrays hold before its count is known; tests force the second round with it).
 */
#ifndef NTLINK_AMD_H
#define NTLINK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTL_OK 0
#define NTL_EINVAL (-1)   /* bad argument */
#define NTL_EDEVICE (-2)  /* no usable gfx950 device / HIP runtime error */
#define NTL_ENOMEM (-3)
#define NTL_EINTERNAL (-4) /* an internal invariant failed (reported, never silently ignored) */
#define NTL_ERANGE (-5)    /* ntl_tally_add: an overhang came out negative (the reference asserts) */
#define NTL_EIO (-6)       /* a file could not be created or written completely; ntl_io_errno() has the errno */

typedef struct ntl_ctx ntl_ctx;
typedef struct ntl_batch ntl_batch;
typedef struct ntl_sketch ntl_sketch;
typedef struct ntl_index ntl_index;
typedef struct ntl_mapres ntl_mapres;

/* ---- context ------------------------------------------------------------------------- */

/* Opens device `device` (a HIP ordinal).  Fails with NTL_EDEVICE if there is no GPU: there is
 * no CPU fallback. */
int ntl_ctx_create(int device, ntl_ctx **out);
void ntl_ctx_destroy(ntl_ctx *ctx);
const char *ntl_last_error(const ntl_ctx *ctx);
/* Human-readable device description ("AMD Instinct MI355X gfx950 256 CUs"). */
const char *ntl_ctx_device_name(const ntl_ctx *ctx);
/* Blocks until all work queued on the context's streams has finished.  Also where the failure of work whose handle was
 * destroyed before it finished is reported (see "Asynchrony" below). */
int ntl_ctx_sync(ntl_ctx *ctx);
/* 1 when the window stage of a sketch runs on its own stream beside the previous batch's emit / map kernels (the default),
 * 0 with NTL_PIPELINE=0 (one stream: what a per-kernel profile wants). */
int ntl_ctx_pipelined(const ntl_ctx *ctx);
/* Turns the second stream off / on again at a quiet point (waits for everything queued first).  NTL_EINVAL when the
 * context was created under NTL_PIPELINE=0. */
int ntl_ctx_set_pipeline(ntl_ctx *ctx, int on);

/* Kernel timing with HIP events on the context's stream.  When enabled, every launch of the
 * named kernel groups is bracketed by events; ntl_prof_get returns the accumulated time and
 * launch count since the last ntl_prof_reset.  Names: "sketch_mask", "sketch_emit", "index",
 * "probe", "map", "compact", "format".  "hipMalloc" (always counted, HOST milliseconds): the misses of the context's cache of
 * device blocks. */
int ntl_prof_enable(ntl_ctx *ctx, int on);
int ntl_prof_reset(ntl_ctx *ctx);
int ntl_prof_get(ntl_ctx *ctx, const char *name, double *total_ms, uint64_t *launches);

/* ---- sequences ------------------------------------------------------------------------ */

/* Hands nseq sequences over: sequence i is the ASCII bytes seqs[offsets[i] .. offsets[i+1]).
 * The bytes are copied to the device as they are and packed there to 2 bits/base plus a table of
 * ACGT runs; the result stays on the device until ntl_batch_destroy and the caller's arrays are
 * free again when the call returns.  Replaces: SeqReader -> NtHash input (ntLink:199,223). */
int ntl_batch_create(ntl_ctx *ctx, const char *seqs, const uint64_t *offsets, uint64_t nseq,
                     ntl_batch **out);
void ntl_batch_destroy(ntl_batch *b);
/* The same from bases that are already in the device's layout -- what ntl_fastx_copy_packed + ntl_fastx_runs produce while
 * they parse: packed[ntl_packed_words(bases)] holds 2 bits per base (A/a 0, C/c 1, G/g 2, T/t 3, anything else 0), sixteen per
 * word, base b of the batch at bit position 2 * (16 + b) (sixteen bases of zero padding in front, 4096 behind); offsets[0] = 0;
 * sequence i has the maximal ACGT/acgt runs seq_run_first[i] .. seq_run_first[i+1], run r = run_len[r] bases from offset
 * run_start[r] of its sequence.  A quarter of the bytes of ntl_batch_create cross PCIe and no pack kernel runs. */
uint64_t ntl_packed_words(uint64_t bases);
int ntl_batch_create_packed(ntl_ctx *ctx, const uint32_t *packed, const uint64_t *offsets, uint64_t nseq,
                            const uint32_t *seq_run_first, const uint32_t *run_start, const uint32_t *run_len, uint64_t nruns,
                            ntl_batch **out);
/* The same from a packed stream whose sequences need not be contiguous -- what the one-pass reader (ntl_fastx_next_span /
 * _parse_span / _copy_span) produces: sequence i = lengths[i] bases from base position positions[i] of the stream (in order,
 * non-overlapping; bit position 2 * (16 + positions[i] + j) for its base j), span_positions = positions the stream spans;
 * packed holds ntl_packed_words(span_positions) words.  The words between sequences are never interpreted. */
int ntl_batch_create_packed_at(ntl_ctx *ctx, const uint32_t *packed, uint64_t span_positions, const uint64_t *positions,
                               const uint32_t *lengths, uint64_t nseq, const uint32_t *seq_run_first, const uint32_t *run_start,
                               const uint32_t *run_len, uint64_t nruns, ntl_batch **out);
uint64_t ntl_batch_nseq(const ntl_batch *b);
uint64_t ntl_batch_bases(const ntl_batch *b);

/* Page-locked host memory for the seqs of ntl_batch_create (one DMA instead of a staged copy);
 * the FASTA/FASTQ reader below can parse straight into it.  Optional: any host pointer works.
 * Blocks of 4 MB and more are anonymous mappings on 2-MB boundaries (transparent huge pages where the host grants them) registered
 * with the runtime -- a fifth of hipHostMalloc's time for the same DMA rate; free them with ntl_host_free only. */
int ntl_host_alloc(ntl_ctx *ctx, uint64_t bytes, void **out);
void ntl_host_free(ntl_ctx *ctx, void *p);

/* ---- synthetic workloads (bench / test harness support; no counterpart in the reference) ---- */

/* BASELINE.json's configurations are synthetic (i.i.d. ACGT genome cut into contigs,
 * ONT/HiFi-like reads with substitution / insertion / deletion errors).  These entry points build such
 * batches directly in HBM, in the layout ntl_batch_create leaves, so that a 3 Gbp assembly and 90
 * Gbases of reads need neither host generation nor PCIe.  All sequences are pure ACGT.
 *   ntl_synth_genome: nseq sequences of len[i] uniform random bases (deterministic in seed).
 *   ntl_synth_slices: sequence i = out_len[i] bases read from source sequence src_seq[i] starting at
 *       src_start[i]; reverse[i] != 0 takes the reverse complement of the slice; sub/ins/del are
 *       per-base event probabilities (0,0,0 = exact copy: contigs).  With errors the slice may consume up
 *       to out_len + out_len/8 + 64 source bases, which must fit the source sequence.
 *   ntl_batch_download: the bases back as ASCII (seqs[offsets[i]..offsets[i+1]), offsets[nseq+1]) --
 *       what the tests hand to the oracle.  Only for pure-ACGT batches. */
int ntl_synth_genome(ntl_ctx *ctx, uint64_t seed, const uint32_t *len, uint64_t nseq, ntl_batch **out);
int ntl_synth_slices(ntl_ctx *ctx, const ntl_batch *src, uint64_t seed, uint64_t n, const uint32_t *src_seq,
                     const uint32_t *src_start, const uint32_t *out_len, const uint8_t *reverse,
                     double sub, double ins, double del, ntl_batch **out);
int ntl_batch_download(const ntl_batch *b, char *seqs, uint64_t *offsets);

/* ---- sketch ---------------------------------------------------------------------------- */

/* Asynchrony.  ntl_sketch_run[_indexed] and ntl_map_run only QUEUE device work and return; no size comes back to the host
 * inside them.  The handles they return are completed on first use: any accessor that needs a count or a record
 * (ntl_sketch_count / _download / _redo_strips, ntl_mapres_n_* / _download, ntl_index_build on a contig sketch) waits for
 * the device then, and ntl_sketch_wait / ntl_mapres_wait do only that and return the status.  A caller that queues batch
 * i+1 before it asks for the results of batch i lets the window kernels of i+1 run beside the lookup and map kernels of i
 * (the reference's pipe between `indexlr` and `ntlink_pair.py` overlaps the same two stages, ntLink:221-225).  Inputs may be
 * destroyed as soon as the call that took them has returned: the library keeps what it still needs by reference count -- the
 * batch a sketch was made from, the sketch a map result was made from, and the INDEX both were queued against (their kernels
 * read it, and a sketch that overflowed its record array is made again from it when its count is asked for); ntl_index_destroy
 * and the other destroy calls only drop the caller's reference.  Destroying a handle that was never asked for anything is
 * allowed; if its work then fails, the next ntl_ctx_sync reports it.
 * Limits: a context holds 512 page-locked result slots, one per PENDING sketch or map result (queued, not yet completed by an
 * accessor or a wait; handles destroyed while pending count until their work has run).  A call that finds none waits for the
 * oldest destroyed-while-pending handle and otherwise fails with NTL_EDEVICE; the library itself never runs more than eight
 * sketches ahead of the device.  COMPLETED handles hold no slot: any number of them may stay alive. */

/* Computes the (k,w) minimizers of every sequence of the batch on the device; the result stays
 * in device memory.  Replaces `indexlr --long --pos --strand -k K -w W` (ntLink:199,223):
 * ntHash canonical hash for the window minimum, second hash as the emitted value, window over
 * valid k-mers, rightmost minimum on ties. */
int ntl_sketch_run(ntl_ctx *ctx, const ntl_batch *b, int k, int w, ntl_sketch **out);
/* The same sketch made FOR one contig index (read batches of the pair stage): every minimizer is looked up in `ix` while it is
 * emitted, and ntl_map_run(ix, this sketch) skips its own lookup pass over the 16-byte records -- per-read lookup of
 * bin/ntlink_pair.py:364-367 fused into the emitter of `indexlr`.  Minimizers and mappings are those of the two-call form;
 * the index may be destroyed once this call has returned (see "Asynchrony"). */
int ntl_sketch_run_indexed(ntl_ctx *ctx, const ntl_batch *b, int k, int w, const ntl_index *ix, ntl_sketch **out);
/* ... and made ONLY to be mapped against `ix` (what the pair stage does with a read batch, whose minimizers the reference
 * never keeps either: bin/ntlink_pair.py:352-367 reads indexlr's lines off a pipe): the 16-byte records are not written.  A
 * minimizer leaves its candidate and its position in the read, the 12 bytes the map kernels read of it (28 with the records).  ntl_sketch_count / _wait / _nseq and ntl_map_run(ix, this sketch) work; ntl_sketch_download with a record column,
 * ntl_index_build, ntl_overlap_filter and a map against another index answer NTL_EINVAL.  Mappings are those of the other forms. */
int ntl_sketch_run_for_map(ntl_ctx *ctx, const ntl_batch *b, int k, int w, const ntl_index *ix, ntl_sketch **out);
/* 0 for a sketch made by ntl_sketch_run_for_map. */
int ntl_sketch_has_records(const ntl_sketch *s);
void ntl_sketch_destroy(ntl_sketch *s);
/* Waits until the sketch is complete; 0 or the error its completion met. */
int ntl_sketch_wait(const ntl_sketch *s);
uint64_t ntl_sketch_nseq(const ntl_sketch *s);
/* Total number of minimizers. */
uint64_t ntl_sketch_count(const ntl_sketch *s);
/* Diagnostics of the window pass: strips (workgroups) it was cut into, and how many of them the 32-bit pass
 * handed to the exact 64-bit pass (k-mers that share a window and the top 32 hash bits: low-complexity
 * sequence).  The result is the same either way. */
uint64_t ntl_sketch_strips(const ntl_sketch *s);
uint64_t ntl_sketch_redo_strips(const ntl_sketch *s);
/* ... and how many strips the threshold pass (71 <= w <= 255: only k-mers with a small key are looked at) handed to the
 * block-minima pass because one of their windows had no such k-mer (about 0.7 % on random sequence). */
uint64_t ntl_sketch_fallback_strips(const ntl_sketch *s);
/* ... and whether the window passes wrote per-strip minimizer LISTS (1: the windows ntLink runs with, 94 <= w <= 255, k <= 64) or
 * the bitmask of one bit per base (0: every other window; NTL_SKETCH_LISTS=0; a sketch whose lists ran out of room -- low-complexity
 * sequence throughout -- and was made again).  The result is the same either way. */
int ntl_sketch_from_lists(const ntl_sketch *s);
/* mx_off[nseq+1]: minimizers of sequence i are [mx_off[i], mx_off[i+1]); hash/pos/strand hold
 * ntl_sketch_count() entries (the three fields `indexlr` prints as H:pos:strand). */
int ntl_sketch_download(const ntl_sketch *s, uint64_t *mx_off, uint64_t *hash, uint32_t *pos,
                        uint8_t *strand);
/* Builds a device-resident sketch from host arrays (the text TSV path of the reference:
 * `ntlink_pair.py -m contigs.tsv FILES`, bin/ntlink_pair.py:195-207,355-378). */
int ntl_sketch_from_host(ntl_ctx *ctx, uint64_t nseq, const uint64_t *mx_off, const uint64_t *hash,
                         const uint32_t *pos, const uint8_t *strand, ntl_sketch **out);

/* ---- overlap-stage consumer (k15 w5 sketch) ---------------------------------------------------- */

/* The minimizers the overlap stage keeps per sequence: read_minimizer_line / read_minimizers of
 * bin/ntlink_overlap_sequences.py:145-190 (ntLink:243-251 pipes `indexlr --long --pos -k 15 -w 5` into it).  Sequence i has the
 * valid regions region_start[r] .. region_end[r] (inclusive positions, is_in_valid_region :138-143) for r in
 * region_off[i] .. region_off[i+1]; a sequence without regions (a name that is not in valid_mx_positions) keeps nothing.  Of the
 * minimizers inside a region, every hash that occurs more than once in the sequence is dropped entirely.  The result is a new
 * device-resident sketch with the same sequences and the kept minimizers in their original order (ntl_sketch_download gives
 * the arrays); the input sketch is left alone. */
int ntl_overlap_filter(ntl_ctx *ctx, const ntl_sketch *s, const uint64_t *region_off, const uint32_t *region_start,
                       const uint32_t *region_end, ntl_sketch **out);

/* ---- contig index ---------------------------------------------------------------------- */

/* Builds the minimizer -> (contig, position, strand) table from the contig sketch; a hash that
 * occurs more than once anywhere is dropped entirely (bin/ntlink_pair.py:204-209).
 * ctg_len[n_ctg] are the contig lengths (bin/ntlink_utils.py:65-73), n_ctg == sketch nseq. */
int ntl_index_build(ntl_ctx *ctx, const ntl_sketch *contigs, const uint32_t *ctg_len, uint32_t n_ctg,
                    ntl_index **out);
void ntl_index_destroy(ntl_index *ix);
/* Number of minimizers kept (unique hashes). */
uint64_t ntl_index_size(const ntl_index *ix);

/* ---- mapping ---------------------------------------------------------------------------- */

typedef struct {
    int32_t k;             /* -k */
    int32_t z;             /* -z minimum contig length (ntLink passes 1000) */
    double x;              /* -x fudge factor; 0 = compare with the read length only */
    int32_t sensitive;     /* --sensitive */
    int32_t repeat_filter; /* --repeat-filter */
} ntl_map_params;

/* One accepted (read, contig) mapping = one line of <prefix>.verbose_mapping.tsv
 * (bin/ntlink_pair.py:382-388); its hits are hits[hit_off .. hit_off + n_hits) in read order. */
typedef struct {
    uint32_t read, ctg, n_hits, pad;
    uint64_t hit_off;
} ntl_mapping;

/* ctg_pos:ctg_strand_read_pos:read_strand (bin/ntlink_pair.py:308-313) */
typedef struct {
    uint32_t ctg_pos, read_pos;
    uint8_t ctg_strand, read_strand, pad[2];
} ntl_hit;

/* One line of <prefix>.paf (bin/ntlink_paf_output.py:123-135); col 11 = t_end - t_start. */
typedef struct {
    uint32_t read, ctg;
    uint32_t q_start, q_end, t_start, t_end;
    uint32_t n_hits;
    uint32_t strand;
} ntl_paf;

/* Maps every read of the read sketch: index lookup, optional repeat filter, contig length and
 * span filters, run grouping, subsumption, accepted contigs, PAF blocks.  read_len[nreads] is
 * the `--len` column.  Results stay on the device until downloaded. */
int ntl_map_run(ntl_ctx *ctx, const ntl_index *ix, const ntl_sketch *reads, const uint32_t *read_len,
                const ntl_map_params *params, ntl_mapres **out);
void ntl_mapres_destroy(ntl_mapres *r);
/* Waits until the result is complete; 0 or the error its completion met (NTL_EINTERNAL: the reference's assertion that every
 * accepted contig appears once per read, bin/ntlink_utils.py:262-266, failed). */
int ntl_mapres_wait(const ntl_mapres *r);
uint64_t ntl_mapres_n_mappings(const ntl_mapres *r);
uint64_t ntl_mapres_n_hits(const ntl_mapres *r);
uint64_t ntl_mapres_n_pafs(const ntl_mapres *r);
/* Number of read minimizers found in the index (before any filter): the hit fraction h. */
uint64_t ntl_mapres_n_index_hits(const ntl_mapres *r);
int ntl_mapres_download(const ntl_mapres *r, ntl_mapping *maps, ntl_hit *hits, ntl_paf *pafs);

/* ---- host-side native I/O (no GPU involved) ------------------------------------------------ */

/* FASTA/FASTQ(.gz) reader = `gzip -cd -f FILE | SeqReader` of the reference's pipe
 * (ntLink:113-117,222-223); record semantics of bin/read_fasta.py:6-46.  path "-" = stdin. */
typedef struct ntl_fastx ntl_fastx;
int ntl_fastx_open(const char *path, ntl_fastx **out);
void ntl_fastx_close(ntl_fastx *r);
const char *ntl_fastx_error(const ntl_fastx *r);
/* Collects records until about max_bases bases are held (0 = whole input); *nseq == 0 at the end.
 * Plain regular files are memory-mapped and parsed by several threads (NTL_IO_THREADS, default
 * min(cores, 32)) over byte ranges cut at record boundaries; gzip and stdin go through zlib on one.
 * ntl_fastx_sizes + ntl_fastx_copy gather the batch into caller-allocated arrays (offsets and
 * name_offsets: nseq + 1 entries): sequence i is seqs[offsets[i]..offsets[i+1]), its id
 * names[name_offsets[i]..name_offsets[i+1]).  The pointer accessors return a contiguous copy held
 * by the reader, valid until the next call. */
int ntl_fastx_next(ntl_fastx *r, uint64_t max_bases, uint64_t *nseq);
/* A reader over the records of a plain (uncompressed, regular) file whose FIRST byte lies in [lo, hi) -- hi = 0: to the
 * end -- so that readers opened on [a, b) and [b, c) together see every record once, in order.  NTL_EINVAL for inputs
 * that cannot be cut (gzip, pipes).  ntl_fastx_range reports the bytes [*lo, *hi) the reader really covers (cut at
 * record starts).  The multi-GPU driver gives every rank its own range of the concatenated read files (ntLink:222). */
int ntl_fastx_open_range(const char *path, uint64_t lo, uint64_t hi, ntl_fastx **out);
void ntl_fastx_range(const ntl_fastx *r, uint64_t *lo, uint64_t *hi);
void ntl_fastx_sizes(const ntl_fastx *r, uint64_t *nseq, uint64_t *bases, uint64_t *name_bytes);
int ntl_fastx_copy(const ntl_fastx *r, char *seqs, uint64_t *offsets, char *names, uint64_t *name_offsets);
/* The current batch in the layout of ntl_batch_create_packed instead of ASCII: packed[ntl_packed_words(bases)], offsets and names
 * as above, *nruns = number of ACGT runs; ntl_fastx_runs then fills seq_run_first[nseq + 1], run_start[nruns], run_len[nruns]. */
int ntl_fastx_copy_packed(ntl_fastx *r, uint32_t *packed, uint64_t *offsets, char *names, uint64_t *name_offsets, uint64_t *nruns);
int ntl_fastx_runs(const ntl_fastx *r, uint32_t *seq_run_first, uint32_t *run_start, uint32_t *run_len);
/* One pass instead of two (count, then parse into place): ntl_fastx_next_span cuts the next span of about max_bases bases
 * without reading it (*span_bytes = 0 at the end of the input; *packed_words = words the packed array must hold: every parser
 * thread's range gets room for as many bases as it has bytes -- half as many for FASTQ -- rounded up to whole words);
 * ntl_fastx_parse_span packs the bases into place and collects records, ids and ACGT runs per thread; ntl_fastx_copy_span
 * hands them over (positions[nseq] = first base of every sequence in the packed stream, lengths[nseq], ids, the run table of
 * ntl_fastx_runs, *span_positions for ntl_batch_create_packed_at).  ntl_fastx_parse_span returns NTL_ERANGE when the span cannot
 * be read this way (a range that ends inside a wrapped FASTQ quality section, qualities shorter than their bases): nothing was
 * consumed and ntl_fastx_next reads the same records the two-pass way. */
int ntl_fastx_next_span(ntl_fastx *r, uint64_t max_bases, uint64_t *span_bytes, uint64_t *packed_words);
int ntl_fastx_parse_span(ntl_fastx *r, uint32_t *packed, uint64_t *nseq, uint64_t *bases, uint64_t *name_bytes, uint64_t *nruns);
int ntl_fastx_copy_span(const ntl_fastx *r, uint64_t *positions, uint32_t *lengths, char *names, uint64_t *name_offsets,
                        uint32_t *seq_run_first, uint32_t *run_start, uint32_t *run_len, uint64_t *span_positions);
const char *ntl_fastx_seqs(ntl_fastx *r);
const uint64_t *ntl_fastx_offsets(ntl_fastx *r);
const char *ntl_fastx_names(ntl_fastx *r);
const uint64_t *ntl_fastx_name_offsets(ntl_fastx *r);

/* Text emitters, written to file descriptor fd.  Names are concatenated ids + offsets[n+1].
 * ntl_write_indexlr: `id\t[len\t]H:pos:strand ...` (ntLink:199,223); lengths == NULL omits --len,
 *                    strand == NULL prints `H:pos` (the `--pos`-only form of ntLink:244,249).
 * ntl_write_verbose: <prefix>.verbose_mapping.tsv (bin/ntlink_pair.py:308-313,382-388).
 * ntl_write_paf:     <prefix>.paf (bin/ntlink_paf_output.py:131-135). */
int ntl_write_indexlr(int fd, uint64_t nseq, const char *names, const uint64_t *name_off, const uint32_t *lengths,
                      const uint64_t *mx_off, const uint64_t *hash, const uint32_t *pos, const uint8_t *strand);
int ntl_write_verbose(int fd, const ntl_mapping *maps, uint64_t n_maps, const ntl_hit *hits,
                      const char *read_names, const uint64_t *read_name_off,
                      const char *ctg_names, const uint64_t *ctg_name_off);
int ntl_write_paf(int fd, const ntl_paf *pafs, uint64_t n, const char *read_names, const uint64_t *read_name_off,
                  const uint32_t *read_len, const char *ctg_names, const uint64_t *ctg_name_off, const uint32_t *ctg_len);

/* indexlr TSV parser: the text input of operator B2 (`id\t[len\t]H:pos:strand ...`, split the way
 * bin/ntlink_pair.py:197-207,355-378 split it), several threads over blocks of whole lines.  path "-"
 * = stdin.  ntl_tsv_next reads about max_bytes of text (0 = all of it) and counts; ntl_tsv_copy
 * fills caller-allocated arrays (name_off / mx_off: nrec + 1 entries, lengths may be NULL): record i
 * has the id names[name_off[i]..name_off[i+1]) and the minimizers mx_off[i]..mx_off[i+1]; strand 1 = '+'.
 * A malformed token fails the call (the reference raises ValueError there).  The arrays are what
 * ntl_sketch_from_host takes.  with_len is a flag word: bit 0 = the `--len` column is present, bit 1 = the tokens are
 * `H:pos` (indexlr --pos without --strand, the overlap stage's input, ntLink:244,249; strand[] is then filled with 1). */
typedef struct ntl_tsv ntl_tsv;
int ntl_tsv_open(const char *path, int with_len, ntl_tsv **out);
void ntl_tsv_close(ntl_tsv *r);
const char *ntl_tsv_error(const ntl_tsv *r);
int ntl_tsv_next(ntl_tsv *r, uint64_t max_bytes, uint64_t *nrec);
void ntl_tsv_sizes(const ntl_tsv *r, uint64_t *nrec, uint64_t *nmx, uint64_t *name_bytes);
int ntl_tsv_copy(const ntl_tsv *r, char *names, uint64_t *name_off, uint32_t *lengths, uint64_t *mx_off,
                 uint64_t *hash, uint32_t *pos, uint8_t *strand);

/* ---- the text of the mapping outputs, made on the device ------------------------------------
 * Replaces the formatting loops of bin/ntlink_pair.py:308-313,382-388 (<prefix>.verbose_mapping.tsv) and
 * bin/ntlink_paf_output.py:131-135 (<prefix>.paf): the lines are laid out and written by kernels from the dense records of a map
 * result and two name tables that live on the device; what crosses PCIe is the text itself, the mappings, and -- all the pair tally
 * needs of the hits (bin/ntlink_pair.py:394-406) -- the first and the last hit of every mapping.  The bytes are those of
 * ntl_write_verbose / ntl_write_paf on the same records. */
typedef struct ntl_names ntl_names;
typedef struct ntl_text ntl_text;
/* Name i is names[off[i] .. off[i+1]); len (may be NULL) the sequences' lengths: both tables of ntl_mapres_format need them
 * (PAF columns 2 and 7).  The arrays are copied; they are the caller's again when the call returns. */
int ntl_names_create(ntl_ctx *ctx, const char *names, const uint64_t *off, const uint32_t *len, uint64_t n, ntl_names **out);
void ntl_names_destroy(ntl_names *t);
/* Completes the result (waits for it), then formats: verbose != 0 the lines of .verbose_mapping.tsv, paf != 0 those of .paf.
 * One more host wait inside (the byte totals size the text arrays).  NTL_ERANGE for more than 4 GB of text in one batch.
 * The text keeps what it needs: the result may be destroyed once this call has returned. */
int ntl_mapres_format(const ntl_mapres *r, const ntl_names *reads, const ntl_names *contigs, int verbose, int paf, ntl_text **out);
void ntl_text_sizes(const ntl_text *t, uint64_t *verbose_bytes, uint64_t *paf_bytes, uint64_t *n_mappings);
/* Any destination may be NULL.  ends: 2 * n_mappings hits, {first, last} per mapping (ntl_tally_add_ends). */
int ntl_text_download(const ntl_text *t, char *verbose, char *paf, ntl_mapping *maps, ntl_hit *ends);
void ntl_text_destroy(ntl_text *t);
/* n bytes at the end of fd (O_APPEND or not: the position is taken once), written by the I/O worker pool in parallel pieces. */
int ntl_write_blob(int fd, const char *p, uint64_t n);

/* ---- host-side pair tally (no GPU involved) ------------------------------------------------- */

/* The contig-pair bookkeeping of bin/ntlink_pair.py (tally_pairs_from_mappings :416-435, add_pair
 * :315-334, calculate_pair_info :222-239, calculate_gap_size :157-187, normalize_pair :213-219)
 * over the records of ntl_map_run.  Pairs are keyed by contig NAME (byte order decides which
 * contig comes first); f = the -f chain-length threshold.  ntl_tally_add takes one batch of reads
 * in order (read_len indexed by ntl_mapping.read) and returns NTL_ERANGE where the reference
 * raises "Gap distance estimation less than 0".  ntl_tally_export lists every pair in order of
 * first appearance: contig indices, orientations (1 = '+'), anchor count and
 * gaps[gap_off[i] .. gap_off[i+1]) in read order (arrays sized from npairs / ngaps). */
typedef struct ntl_tally ntl_tally;
int ntl_tally_create(const char *ctg_names, const uint64_t *ctg_name_off, const uint32_t *ctg_len, uint64_t n_ctg,
                     int k, int f, ntl_tally **out);
void ntl_tally_destroy(ntl_tally *t);
int ntl_tally_add(ntl_tally *t, const ntl_mapping *maps, uint64_t n_maps, const ntl_hit *hits, const uint32_t *read_len);
/* The same from the first and last hit of every mapping alone (ends[2 i], ends[2 i + 1]: ntl_text_download). */
int ntl_tally_add_ends(ntl_tally *t, const ntl_mapping *maps, uint64_t n_maps, const ntl_hit *ends, const uint32_t *read_len);
uint64_t ntl_tally_npairs(const ntl_tally *t);
uint64_t ntl_tally_ngaps(const ntl_tally *t);
int ntl_tally_export(const ntl_tally *t, uint32_t *src, uint8_t *src_ori, uint32_t *tgt, uint8_t *tgt_ori,
                     uint32_t *anchor, uint64_t *gap_off, int64_t *gaps);
/* Appends an export of ANOTHER tally over the same contigs that covers later reads (pairs first seen there go behind the
 * known ones, gaps behind the gaps of the same pair, anchors add up): what one tally would hold after both read ranges.
 * The multi-GPU driver tallies per rank and merges in rank order on rank 0. */
int ntl_tally_merge(ntl_tally *t, uint64_t npairs, const uint32_t *src, const uint8_t *src_ori, const uint32_t *tgt,
                    const uint8_t *tgt_ori, const uint32_t *anchor, const uint64_t *gap_off, const int64_t *gaps);

/* The two small files behind the tally: <prefix>.pairs.tsv (write_pairs, bin/ntlink_pair.py:490-496; pairs_path may be NULL) and
 * <prefix>.n<n>.scaffold.dot (build_scaffold_graph :263-305, filter_graph_global :498-506 with min_n = n, print_directed_graph
 * :133-155; dot_path may be NULL) from the pairs that pass filter_pairs_distances (:247-255) and filter_weak_anchor_pairs
 * (:241-244, anchor >= a).  *n_kept (may be NULL): the pairs that passed.  Each file is written beside its place and renamed when it
 * is complete (the reference leaves no partial output behind, bin/ntlink_pair.py:608-613); NTL_EIO when a file cannot be created,
 * written or closed, with the errno in ntl_io_errno(). */
int ntl_tally_write(const ntl_tally *t, int a, int min_n, const char *pairs_path, const char *dot_path, uint64_t *n_kept);
/* errno of the calling thread's last call that returned NTL_EIO (0: none). */
int ntl_io_errno(void);

/* ---- liftover of the verbose mappings (no GPU involved) -------------------------------------- */

/* <prefix>.verbose_mapping.tsv moved to the coordinates of the scaffolds an AGP describes, file to file: the work of
 * bin/ntlink_liftover_mappings.py (liftover_ctg_mappings :61-87, print_adjusted_mappings :89-121, liftover_mappings
 * :125-143; called between rounds, ntLink_rounds:124-125).  The AGP's sequence lines (component type not N / P, read_agp
 * :39-50) come as arrays, one entry per contig id (the last line of an id wins, as in the reference's dict): contig id,
 * path id, scaf_start, ctg_start, ctg_end (1-based, inclusive) and the orientation byte.  Per line: a contig that is not in
 * the AGP keeps its name and loses its mappings; mappings outside [ctg_start-1, ctg_end-k] are dropped; `+` / `-`
 * components of a path with another name are shifted / mirrored (strand flipped).  Per read (consecutive lines with one
 * read id): lines are grouped by path id, paths strictly between two occurrences of a path's first run and a later run
 * are subsumed, the surviving lines of a path are concatenated and printed when their contig positions are strictly
 * monotonic.  A malformed line fails the call (the reference raises) and removes the output file.  lines_in / lines_out
 * may be NULL. */
int ntl_liftover(const char *mappings_path, const char *out_path, int k, uint64_t n_agp, const char *ctg_ids,
                 const uint64_t *ctg_id_off, const char *path_ids, const uint64_t *path_id_off, const int64_t *scaf_start,
                 const int64_t *ctg_start, const int64_t *ctg_end, const char *orientation, uint64_t *lines_in,
                 uint64_t *lines_out);

#ifdef __cplusplus
}
#endif
#endif
