// Small RAII wrapper over rocFFT plans (used by stolt.hip and phaseshift.hip).
#pragma once
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include "common.h"
#include <rocfft/rocfft.h>

#define IMPDAR_FFT_CHECK(expr)                                                          \
    do {                                                                                \
        rocfft_status _s = (expr);                                                      \
        if (_s != rocfft_status_success) {                                              \
            impdar_set_error("%s:%d: %s -> rocfft status %d", __FILE__, __LINE__, #expr, (int)_s); \
            return IMPDAR_ERR_FFT;                                                      \
        }                                                                               \
    } while (0)

int impdar_fft_global_setup();   // rocfft_setup once per process
// rocFFT 1.0.36 creates plans safely only one at a time (see impdar_parallel_plans): every rocfft_plan_create of the
// library runs under this process-wide lock
std::mutex &impdar_fft_plan_mutex();

struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    DevBuf work;
    ~FftPlan() { release(); }
    void release()
    {
        if (info) rocfft_execution_info_destroy(info);
        if (plan) rocfft_plan_destroy(plan);
        info = nullptr;
        plan = nullptr;
        work.release();
    }
    // 1-D batched transform.  stride/dist in elements of the respective type.
    int create(rocfft_transform_type type, bool dbl, bool inplace, size_t length, size_t batch,
               rocfft_array_type in_t, rocfft_array_type out_t, size_t in_stride, size_t in_dist,
               size_t out_stride, size_t out_dist, double scale, hipStream_t stream)
    {
        release();
        int rc = impdar_fft_global_setup();
        if (rc) return rc;
        rocfft_plan_description desc = nullptr;
        IMPDAR_FFT_CHECK(rocfft_plan_description_create(&desc));
        rocfft_status s = rocfft_plan_description_set_data_layout(desc, in_t, out_t, nullptr, nullptr, 1, &in_stride,
                                                                  in_dist, 1, &out_stride, out_dist);
        if (s == rocfft_status_success && scale != 1.0) s = rocfft_plan_description_set_scale_factor(desc, scale);
        std::lock_guard<std::mutex> plan_lock(impdar_fft_plan_mutex());
        impdar_trace("rocfft_plan_create 1-D type %d length %zu batch %zu strides %zu/%zu: start", (int)type, length, batch, in_stride, out_stride);
        if (s == rocfft_status_success)
            s = rocfft_plan_create(&plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace, type,
                                   dbl ? rocfft_precision_double : rocfft_precision_single, 1, &length, batch, desc);
        impdar_trace("rocfft_plan_create 1-D type %d length %zu batch %zu: done", (int)type, length, batch);
        rocfft_plan_description_destroy(desc);
        if (s != rocfft_status_success) {
            impdar_set_error("rocFFT plan creation failed (length %zu, batch %zu, status %d)", length, batch, (int)s);
            return IMPDAR_ERR_FFT;
        }
        IMPDAR_FFT_CHECK(rocfft_execution_info_create(&info));
        IMPDAR_FFT_CHECK(rocfft_execution_info_set_stream(info, stream));
        size_t wb = 0;
        IMPDAR_FFT_CHECK(rocfft_plan_get_work_buffer_size(plan, &wb));
        if (wb) {
            IMPDAR_HIP_CHECK(work.ensure(wb));
            IMPDAR_FFT_CHECK(rocfft_execution_info_set_work_buffer(info, work.p, wb));
        }
        return IMPDAR_OK;
    }
    // 2-D transform of one array: len0 along the contiguous axis, len1 rows `in_row` / `out_row` elements apart.
    // in_col / out_col != 0: that side is stored with the SECOND dimension contiguous instead (element (i0, i1) at
    // i0 * col + i1): handing rocFFT's own transposed intermediate to the next kernel saves it a transpose pass.
    int create2d(rocfft_transform_type type, bool dbl, bool inplace, size_t len0, size_t len1, rocfft_array_type in_t,
                 rocfft_array_type out_t, size_t in_row, size_t out_row, double scale, hipStream_t stream,
                 size_t in_col = 0, size_t out_col = 0, size_t in_dist = 0, size_t out_dist = 0)
    {
        release();
        int rc = impdar_fft_global_setup();
        if (rc) return rc;
        rocfft_plan_description desc = nullptr;
        IMPDAR_FFT_CHECK(rocfft_plan_description_create(&desc));
        size_t is[2] = {1, in_row}, os[2] = {1, out_row};
        if (in_col) { is[0] = in_col; is[1] = 1; }
        if (out_col) { os[0] = out_col; os[1] = 1; }
        rocfft_status s = rocfft_plan_description_set_data_layout(desc, in_t, out_t, nullptr, nullptr, 2, is,
                                                                  in_dist ? in_dist : in_row * len1, 2, os,
                                                                  out_dist ? out_dist : out_row * len1);
        if (s == rocfft_status_success && scale != 1.0) s = rocfft_plan_description_set_scale_factor(desc, scale);
        size_t lengths[2] = {len0, len1};
        std::lock_guard<std::mutex> plan_lock(impdar_fft_plan_mutex());
        impdar_trace("rocfft_plan_create 2-D type %d %zu x %zu: start", (int)type, len0, len1);
        if (s == rocfft_status_success)
            s = rocfft_plan_create(&plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace, type,
                                   dbl ? rocfft_precision_double : rocfft_precision_single, 2, lengths, 1, desc);
        impdar_trace("rocfft_plan_create 2-D type %d %zu x %zu: done", (int)type, len0, len1);
        rocfft_plan_description_destroy(desc);
        if (s != rocfft_status_success) {
            impdar_set_error("rocFFT 2-D plan creation failed (%zu x %zu, status %d)", len0, len1, (int)s);
            return IMPDAR_ERR_FFT;
        }
        IMPDAR_FFT_CHECK(rocfft_execution_info_create(&info));
        IMPDAR_FFT_CHECK(rocfft_execution_info_set_stream(info, stream));
        size_t wb = 0;
        IMPDAR_FFT_CHECK(rocfft_plan_get_work_buffer_size(plan, &wb));
        if (wb) {
            IMPDAR_HIP_CHECK(work.ensure(wb));
            IMPDAR_FFT_CHECK(rocfft_execution_info_set_work_buffer(info, work.p, wb));
        }
        return IMPDAR_OK;
    }
    int exec(void *in, void *out)
    {
        void *ib[1] = {in};
        void *ob[1] = {out};
        IMPDAR_FFT_CHECK(rocfft_execute(plan, ib, out ? ob : nullptr, info));
        return IMPDAR_OK;
    }
};

// The rocFFT plans a call still lacks, created ONE AFTER THE OTHER (each under the process-wide lock of FftPlan::create).  Rounds 3-4 created them
// side by side on threads (a plan of a new size costs 0.25-2 s on a machine whose kernel cache is cold): with rocFFT
// 1.0.36 that is not safe -- three plans of length 8192 created concurrently made the first kernels after them die with a
// GPU memory access fault in 3 of 13 fresh processes, never in 16 with the plans in turn (profiles/r05_first_call.txt).
// Returns the first failure.
static int impdar_parallel_plans(int device, std::vector<std::function<int()>> makers)
{
    (void)device;           // (the lock is taken per plan: FftPlan::create)
    for (auto &m : makers) {
        const int rc = m();
        if (rc) return rc;
    }
    return IMPDAR_OK;
}

// edge taper weight, reference mig_python.py:152-156: min(i, n-1-i)/taper clipped to 1
__host__ __device__ static inline double impdar_taper_w(int i, int n, double taper)
{
    const int m = i < n - 1 - i ? i : n - 1 - i;
    double w = (double)m / taper;
    if (w > 1.0) w = 1.0;
    return w;
}
