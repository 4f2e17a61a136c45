#include <hip/hip_runtime.h>
using buf_rsrc = __amdgpu_buffer_rsrc_t;
typedef unsigned u32x4 __attribute__((__vector_size__(16)));
__device__ __forceinline__ float4 buf_load4(buf_rsrc r, int off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__builtin_bit_cast(float, v[0]), __builtin_bit_cast(float, v[1]), __builtin_bit_cast(float, v[2]),
                       __builtin_bit_cast(float, v[3]));
}
__global__ void k(float4* out, const float* in, int n, const int* rows) {
    buf_rsrc ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    int r = rows[threadIdx.x];
    float4 raw[2];
    raw[0] = buf_load4(ri, r >= 0 ? r * 16 : 0x7fffffff);
    out[threadIdx.x] = raw[0];
}
