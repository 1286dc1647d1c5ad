// resample.hip — the sinc resampler of the step in front of the path on the device.
//
// The reference resamples a capture device's rate (44.1 / 48 kHz) to 16 kHz on the host with rubato's SincFixedIn
// (src/input/audio.rs:1007-1095; restated for the host in dsp.cpp, ohw_dsp_resample_sinc).  A recording that is already in
// HBM - or goes there anyway for the log-mel - can be resampled where it lies: output k is the linear blend of two
// 256-tap dot products of the input around position -128 + (k + 1) / ratio with the two nearest of 256 sub-filters, every
// output independent of the others.  One wave per output: a lane takes 4 consecutive taps of both sub-filters (float4
// loads: the input window and a sub-filter are contiguous), the wave reduces.  HBM-/L2-bound by the table reads (2 KB per
// output out of a 256-KB table that stays in L2): 30 s at 48 kHz -> 480 000 outputs, a fraction of a millisecond.
// Same output count and - up to fp32 summation order - the same samples as ohw_dsp_resample_sinc (tests/test_gpu_resample.py).
#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "common.hpp"
#include "model.hpp"

namespace ohw {
extern thread_local std::string g_last_error;                    // engine.hip
void sinc_table(double ratio, std::vector<float>& sincs);      // dsp.cpp
int64_t sinc_plan(int64_t n, double ratio);

constexpr int RS_L = 256, RS_F = 256;

namespace {
template <typename F>
int rs_guard(F&& f) {
  ApiScope api;        // the library's gate (common.hpp): no launch while another thread captures a graph
  try {
    f();
    return OHW_OK;
  } catch (const Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    g_last_error = "host allocation failed";
    return OHW_E_OOM;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return OHW_E_TRANSCRIBE;
  }
}
}  // namespace

__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ in, int64_t n_in, const float* __restrict__ sincs,
                                                             double t_ratio, float* __restrict__ out, int64_t n_out) {
  const int lane = threadIdx.x & 63;
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_out) return;                                   // whole waves leave together
  const double idx = -(double)(RS_L / 2) + (double)(k + 1) * t_ratio;
  const double fl = floor(idx);
  int64_t i0 = (int64_t)fl;
  int s0 = (int)floor((idx - fl) * (double)RS_F);
  if (s0 > RS_F - 1) s0 = RS_F - 1;
  int64_t i1 = i0;
  int s1 = s0 + 1;
  if (s1 >= RS_F) { s1 -= RS_F; i1 += 1; }
  const double scaled = idx * (double)RS_F;
  const float frac = (float)(scaled - floor(scaled));
  auto dot = [&](int64_t first, int sub) {
    const f32x4 h = *(const f32x4*)(sincs + (int64_t)sub * RS_L + lane * 4);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = first + lane * 4 + j;
      const float w = (p >= 0 && p < n_in) ? in[p] : 0.f;     // zeros in front of the recording and behind it
      acc += w * h[j];
    }
    return wave_sum(acc);
  };
  const float p0 = dot(i0, s0), p1 = dot(i1, s1);
  if (lane == 0) out[k] = p0 + frac * (p1 - p0);
}
}  // namespace ohw

struct ohw_resampler {
  int device = 0;
  uint32_t from = 0, to = 0;
  double ratio = 1.0;
  ohw::DevBuf table, in_buf, out_buf;
};

extern "C" {

int ohw_resampler_create(int device, uint32_t from_rate, uint32_t to_rate, ohw_resampler** out) {
  return ohw::rs_guard([&] {
    if (!out || from_rate == 0 || to_rate == 0) throw ohw::Error(OHW_E_INVALID_ARG, "resampler: null or zero rate");
    const double ratio = (double)to_rate / (double)from_rate;
    if (ratio > 16.0 || ratio < 1.0 / 16.0) throw ohw::Error(OHW_E_INVALID_ARG, "resampler: rate ratio outside 1/16 .. 16");
    HIP_CHECK(hipSetDevice(device));
    ohw_resampler* r = new ohw_resampler();
    r->device = device; r->from = from_rate; r->to = to_rate; r->ratio = ratio;
    try {
      if (from_rate != to_rate) {
        std::vector<float> t;
        ohw::sinc_table(ratio, t);
        r->table.alloc(t.size() * 4);
        HIP_CHECK(hipMemcpy(r->table.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
      }
    } catch (...) { delete r; throw; }
    *out = r;
  });
}

void ohw_resampler_free(ohw_resampler* r) { delete r; }

int64_t ohw_resampler_out_len(const ohw_resampler* r, int64_t n) {
  if (!r || n <= 0) return 0;
  return r->from == r->to ? n : ohw::sinc_plan(n, r->ratio);
}

int ohw_resampler_run(ohw_resampler* r, const float* in, int64_t n, int in_on_device, float* out, int64_t out_cap, int out_on_device,
                      void* hip_stream) {
  return ohw::rs_guard([&] {
    if (!r || !in || !out || n <= 0) throw ohw::Error(OHW_E_INVALID_ARG, "resampler: null or empty");
    const int64_t n_out = ohw_resampler_out_len(r, n);
    if (out_cap < n_out) throw ohw::Error(OHW_E_INVALID_ARG, "resampler: output buffer too small (ohw_resampler_out_len)");
    HIP_CHECK(hipSetDevice(r->device));
    hipStream_t s = (hipStream_t)hip_stream;
    const float* din = in;
    if (!in_on_device) {
      if (r->in_buf.bytes < (size_t)n * 4) r->in_buf.alloc((size_t)n * 4);
      HIP_CHECK(hipMemcpyAsync(r->in_buf.p, in, (size_t)n * 4, hipMemcpyHostToDevice, s));
      din = r->in_buf.as<float>();
    }
    float* dout = out;
    if (!out_on_device) {
      if (r->out_buf.bytes < (size_t)n_out * 4) r->out_buf.alloc((size_t)n_out * 4);
      dout = r->out_buf.as<float>();
    }
    if (r->from == r->to) {
      HIP_CHECK(hipMemcpyAsync(dout, din, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    } else if (n_out > 0) {
      const int64_t blocks = (n_out + 3) / 4;
      if (blocks >= ((int64_t)1 << 31)) throw ohw::Error(OHW_E_INVALID_ARG, "resampler: too many output samples for one launch");
      hipLaunchKernelGGL(ohw::resample_sinc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, din, n, r->table.as<float>(), 1.0 / r->ratio, dout, n_out);
      HIP_CHECK(hipGetLastError());
    }
    if (!out_on_device) {
      HIP_CHECK(hipMemcpyAsync(out, dout, (size_t)n_out * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
    } else if (!in_on_device) {
      HIP_CHECK(hipStreamSynchronize(s));          // the caller's host buffer may go away
    }
  });
}

}  // extern "C"
