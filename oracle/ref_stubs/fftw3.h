/* fftw3.h -- stand-in for FFTW 3.3's single-precision API, just the entry points src/oc_fftcc.cpp of the reference
 * calls (:31-42, 59-70, 75-85, 233-243, 378-388), so that file compiles UNMODIFIED into oracle/_ref/liboc_ref.so.
 * TEST INFRASTRUCTURE ONLY.  The transforms are plain DFTs evaluated in double precision and rounded once to float
 * (oracle/ref_stubs/fftw3_stub.cpp): FFTW's own butterfly order is an FFTW internal; what this pins is everything
 * AROUND the transforms -- window fill, means, the float spectrum product, peak search, wrap-around, ZNCC. */
#ifndef OC_REF_STUB_FFTW3_H
#define OC_REF_STUB_FFTW3_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float fftwf_complex[2];
typedef struct oc_stub_fftwf_plan_s* fftwf_plan;

#define FFTW_MEASURE (0U)
#define FFTW_ESTIMATE (1U << 6)

void* fftw_malloc(size_t n);
void fftw_free(void* p);
void* fftwf_malloc(size_t n);
void fftwf_free(void* p);

/* n0 is the slowest dimension, as in FFTW; the half-spectrum has n_last / 2 + 1 entries along the fastest one */
fftwf_plan fftwf_plan_dft_r2c_2d(int n0, int n1, float* in, fftwf_complex* out, unsigned flags);
fftwf_plan fftwf_plan_dft_c2r_2d(int n0, int n1, fftwf_complex* in, float* out, unsigned flags);
fftwf_plan fftwf_plan_dft_r2c_3d(int n0, int n1, int n2, float* in, fftwf_complex* out, unsigned flags);
fftwf_plan fftwf_plan_dft_c2r_3d(int n0, int n1, int n2, fftwf_complex* in, float* out, unsigned flags);
void fftwf_execute(const fftwf_plan plan);
void fftwf_destroy_plan(fftwf_plan plan);

#ifdef __cplusplus
}
#endif

#endif
