/* abea_device.h — structures shared between the host driver (abea_capi.cpp) and the gfx950
 * kernels (abea_kernels.hip).  Internal; the public ABI is include/abea.h. */
#ifndef ABEA_DEVICE_H
#define ABEA_DEVICE_H
#include <stdint.h>
#include "../../include/abea.h"

#define ABEA_W          ABEA_BANDWIDTH   /* 100 cells per band */
#define ABEA_GROUP      32               /* bands per trace group (one uint4 per lane) */
#define ABEA_WAVE       64
#define ABEA_MOVE_LANE  50               /* lane whose trace nibble carries the band-move bit */

/* Per-k-mer emission parameters produced by the align-pre kernel (16 B, read-scaled):
 *   gpm  = scale*level_mean + shift           (float mul, float add; align.c:137-138)
 *   ck   = -0.918938f - level_log_stdv        (align.c:111,113 first two terms)
 *   istd = 1.0 / (double)level_stdv           ((float)((double)(x-gpm)*istd) is the
 *                                              correctly rounded float quotient (x-gpm)/stdv) */
typedef struct __attribute__((aligned(16))) { float gpm; float ck; double istd; } abea_kpar_t;

/* Per-read descriptor built on the host (per-read doubles come from glibc log/exp). */
typedef struct __attribute__((aligned(16))) {
    int64_t read_off;     /* chars  into reads   */
    int64_t event_off;    /* events into events  */
    int64_t pair_off;     /* pairs  into pairs   */
    int64_t kpar_off;     /* abea_kpar_t into the k-mer parameter scratch */
    int64_t evm_off;      /* floats into the event-mean scratch */
    int64_t trace_off;    /* uint4  into the trace scratch (n_groups * 64 uint4) */
    int64_t code_off;     /* uint32 into the traceback-code scratch */
    int64_t kmer_off;     /* abea_index_pair_t into base_to_event_map (optional scaling outputs) */
    int64_t pad64;        /* unused since round 4 (was: the 'M'-state record scratch of the separate scaling kernels) */
    int32_t read_len, n_events, n_kmers, n_groups;
    float   scale, shift;
    int32_t out_idx;      /* index of the read in the caller's n_pairs[] / diag[] */
    int32_t pad;
    double  lp_skip, lp_stay, lp_step, lp_trim;   /* align.c:212-216 */
} abea_read_desc;

/* scaling_single() fused behind the alignment, inside abea_align_kernel (round 4): the wavefront that aligned a read also runs
 * postalign + recalibrate_model + the FAILED_* flags for it (src/f5c.c:736-807, src/align.c:561-773).  Passed by value; b2e ==
 * NULL switches the stage off.  The pair lists need not be materialised for it (pairs_all may be NULL). */
typedef struct {
    const char* reads;                  /* sequences of the launch (desc.read_off) */
    const abea_model_t* model;          /* k-mer model on the device */
    abea_index_pair_t* b2e;             /* base_to_event_map: entry k of a read at b2e[desc.kmer_off + k] */
    abea_scalings_t* sc_io;             /* [out_idx] estimated scalings in, recalibrated shift / scale / var out */
    double* epb;                        /* [out_idx] events_per_base */
    int32_t* flag_io;                   /* [out_idx] FAILED_* bits OR-ed in */
    int32_t* nalign;                    /* [out_idx] n_event_alignment */
    uint8_t* kcnt;                      /* optional: events of every k-mer's map entry (stop - start + 1, 0 = {-1,-1}, 255 = more),
                                           at kcnt[desc.kmer_off + k]: the form in which the host entry takes the map over PCIe */
    double* var_f64;                    /* optional [out_idx]: recalibrate_model's var in double when the read was recalibrated, -1 otherwise:
                                           the host entries set scalings_t.log_var = (float)log(var) from it with glibc (align.c:760) */
    int32_t kmer_size, min_rescale;
    /* round 6: what phase 4 used to re-derive per k-mer and sweep.  krank[desc.kpar_off + k] = the k-mer's rank, left by align-pre
     * (which has it in a register anyway); mterms[3 * rank + {0, 1, 2}] = 1/sd^2, mu/sd^2... exactly: inv_var = 1. / (sd * sd),
     * mu * inv_var, mu * mu * inv_var in double for every model entry — the three normal-equation terms of align.c:697-706 that
     * depend on the model alone, computed once at abea_init with the reference's expressions. */
    const uint32_t* krank;
    const double* mterms;
} abea_fused_scaling;

#endif
