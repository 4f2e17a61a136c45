#include <hip/hip_runtime.h>
using buf_rsrc = __amdgpu_buffer_rsrc_t;
typedef unsigned u32x4 __attribute__((__vector_size__(16)));
__global__ void k1(float4* out, const float* in, int n, const int* rows) {
    buf_rsrc ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    int r = rows[threadIdx.x];
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ri, r * 16, 0, 0);
    out[threadIdx.x] = make_float4(__builtin_bit_cast(float, v[0]), __builtin_bit_cast(float, v[1]), __builtin_bit_cast(float, v[2]), __builtin_bit_cast(float, v[3]));
}
__global__ void k2(u32x4* out, const float* in, int n, const int* rows) {
    buf_rsrc ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    int r = rows[threadIdx.x];
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ri, r * 16, 0, 0);
    out[threadIdx.x] = v;
}
__global__ void k3(float* out, const float* in, int n, const int* rows) {
    buf_rsrc ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    int r = rows[threadIdx.x];
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ri, r * 16, 0, 0);
    float4 f = *reinterpret_cast<float4*>(&v);
    out[threadIdx.x] = f.x + 2 * f.y + 3 * f.z + 4 * f.w;
}
