/*
 * reflectance_filtering_debug.h -- test and benchmark switches of librf_hip.so.
 *
 * NOT part of the drop-in boundary (include/reflectance_filtering.h): nothing here is needed
 * to replace the reference's native calls.  The switches select alternative kernels that the
 * parity tests and the timing tools compare against the default ones; they are process-global,
 * not meant to be flipped while other threads are inside the library, and all default to 0.
 *
 *   "gf_two_kernel"      guided filter: row-sum / column-sum kernel pair for every radius (the
 *                        default fuses stage 2 for every radius 1..128); identical bytes
 *   "jbf_compiler_loop"  joint bilateral: compiler-scheduled tap loop; identical bytes.  (This switch,
 *                        "jbf_lookahead1" and "jbf_stage_only" act on the 64x64-tile kernel, radius <= 52;
 *                        the slab kernel of radius 53..468 has one tap loop and ignores them - its
 *                        independent check is RF_JBF_FORCE_GENERIC, tests/test_gpu_parity.py)
 *   "jbf_tile64_only"    joint bilateral: no strip tiles at the image remainder; identical bytes
 *   "jbf_tune"           joint bilateral: kernel-variant override 1..7 (tools/jbf_tune.py)
 *   "jbf_f32_untiled"    float joint bilateral: one-thread-per-pixel kernel; identical values
 *   "cnn_lds_columns"    CNN: activations pass between layers through LDS columns instead of
 *                        registers; identical values
 *   "gf_seg_rows"        guided filter: rows per stage-1 segment (tools/gf_seg_sweep.py); 0 =
 *                        chosen by the library; identical bytes
 *   "gf_one_stream"      guided filter: the whole batch on the caller's stream (the default runs
 *                        the two halves of a batch on the caller's stream and on a side stream of
 *                        the library, forked / joined with events inside the call); identical bytes
 *   "gf_force_two_streams"  guided filter: fork the side stream for every chunk of two or more
 *                        images (the default decides by batch size and src kind); identical bytes
 *   "gf_guide_cache"     guided filter, experiment: an iterated call keeps the guide's window
 *                        statistics of its first pass in the workspace (36 B per pixel) and later
 *                        passes box-sum only the src quantities; identical bytes, measured slower
 *                        (profiles/r03_gf_guide_cache.md), off by default
 *   "gf_chained"         guided filter, radius 45 / 52: the column walk without a row-walk kernel
 *                        (every 16-column block takes its row sums from its left neighbour through
 *                        tagged slots); identical bytes, measured slower
 *                        (profiles/r04_gf_chained.md), off by default
 *   "gf_no_compact"      guided filter: grey images of a 3-channel src are read and handed from pass
 *                        to pass as three channels (the default keeps them as one byte per pixel in
 *                        the workspace); identical bytes
 *   "jbf_stage_only"     joint bilateral: stage the tile and return WITHOUT WRITING dst
 *                        (tools/jbf_tune.py --stage-only, timing only)
 *   "gf_exp_skip"        guided filter, TIMING ONLY, WRONG RESULTS: bit 0 no stage 1, bit 1 no row
 *                        walk, bit 2 no column walk (tools/gf_c5_exp.py)
 *   "jbf_lookahead1"     joint bilateral: the grey asm tap loop with its LUT gathers one column step
 *                        ahead of their use and a full wait per step (the round-4 form; the default keeps
 *                        them two steps ahead, four gathers in flight across a step); identical bytes
 *   "gf_stagger"         guided filter: staggered two-stream schedule - the stage-1 launches of the
 *                        parts of a chunk are chained by events in (pass, part) order, parts alternating
 *                        between the caller's stream and the side stream, so that one part's stage 1
 *                        always faces the other's walks; identical bytes, measured no faster than the
 *                        default (profiles/r05_c5_overlap.md), off by default
 *   "gf_parts"           ... parts per chunk in that schedule (even, 2..16; 0 = 2)
 *   "gf_s1_cap"          guided filter: stage-1 workgroups per CU (1..3, by a dynamic-LDS pad; 0 =
 *                        whatever fits); identical bytes
 *   "gf_s1_min_wgs"      guided filter: workgroups a stage-1 launch should at least have (chooses the
 *                        rows per segment; 0 = chosen by the library); identical bytes
 *   "gf_s1_legacy_strips"  guided filter: stage-1 strips with a halo of exactly r columns on either side
 *                        (rounds 1-5; the default aligns halo and width to 16 columns and takes a whole
 *                        wave of halo where that costs no strip); identical bytes
 *   "gf_exact"           guided filter, radius 45 / 52, width a multiple of 16: exact-row stage 2 - rows
 *                        whose alpha / beta pass an exactness test take no sequential row walk (block sums
 *                        from stage 1, the rest listed and walked); identical bytes, measured slower
 *                        (profiles/r06_gf_exact.md), off by default
 *   "gf_exact_all_flagged"  ... with every row treated as failing the test (exercises the list path)
 *   "gf_cw_chan_run"     guided filter, colour src, passes of an iterated call that hand their result on as
 *                        planes: n + 1 = the column walk takes an XCD's (block, channel) items in runs of n
 *                        blocks per channel (0 = the library's choice, 64; 1 = channel fastest); identical bytes
 */
#ifndef REFLECTANCE_FILTERING_DEBUG_H
#define REFLECTANCE_FILTERING_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Sets option `name` to `value` (>= 0) and returns its previous value; RF_E_BADARG (-1) for an
 * unknown name. */
int rf_debug_option(const char *name, int value);

/* Measurement aid of bench.py: one wave on `stream` brackets `micros` microseconds of wall time
 * (s_memrealtime, 100 MHz) with the shader-cycle counter (s_memtime), sleeping in between, and
 * writes {shader cycles, 100 MHz ticks} to the two device words at out2.  Launched on a second
 * stream beside a running kernel it reads the clock the chip holds under that kernel's load:
 * MHz = 100 * out2[0] / out2[1]. */
int rf_debug_clock_probe(unsigned long long *out2, int micros, void *stream);

/* The toolchain this library was compiled with (`hipcc --version`, first two lines): the inline-asm
 * hazard audit of tests/test_cabi.py holds for the machine code of that compiler. */
const char *rf_debug_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* REFLECTANCE_FILTERING_DEBUG_H */
