#include <hip/hip_runtime.h>
template <int D> __device__ __forceinline__ int xor_lane(int v) {
    if constexpr (D == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);
    else if constexpr (D == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);
    else if constexpr (D == 4) { int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false); return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false); }
    else if constexpr (D == 8) { int r = __builtin_amdgcn_update_dpp(v, v, 0x108, 0xF, 0x3, false); return __builtin_amdgcn_update_dpp(r, v, 0x118, 0xF, 0xC, false); }
    else if constexpr (D == 16) { auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (threadIdx.x & 16) ? p[0] : p[1]; }
    else { auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (threadIdx.x & 32) ? p[0] : p[1]; }
}
__global__ void k(int* p) {
    int v = p[threadIdx.x];
    p[threadIdx.x] = xor_lane<1>(v) + 3*xor_lane<2>(v) + 5*xor_lane<4>(v) + 7*xor_lane<8>(v) + 11*xor_lane<16>(v) + 13*xor_lane<32>(v);
}
int main() {
    int h[64], *d; for (int i = 0; i < 64; ++i) h[i] = i * 1000;
    hipMalloc(&d, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d); int o[64]; hipMemcpy(o, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { int e = 1000 * ((i^1) + 3*(i^2) + 5*(i^4) + 7*(i^8) + 11*(i^16) + 13*(i^32)); if (o[i] != e) { ++bad; printf("lane %d got %d want %d\n", i, o[i], e); } }
    printf("xor_lane check: %s\n", bad ? "FAILED" : "ok");
    return bad != 0;
}
