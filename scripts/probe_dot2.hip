// probe: chained __builtin_amdgcn_fdot2_f32_bf16 (v_dot2c_f32_bf16) on gfx950: dependent
// back-to-back accumulation vs the same sum with f32 FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void k(const u32x4* a, const u32x4* b, float* o) {
    int i = threadIdx.x;
    u32x4 x = a[i], y = b[i];
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x[j]), __builtin_bit_cast(bf16x2, y[j]), acc, false);
    float acc2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc2) : "v"(x[j]), "v"(y[j]));
    float ref = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ref = fmaf(__uint_as_float(x[j] << 16), __uint_as_float(y[j] << 16), ref);
        ref = fmaf(__uint_as_float(x[j] & 0xffff0000u), __uint_as_float(y[j] & 0xffff0000u), ref);
    }
    o[i] = acc; o[64 + i] = ref; o[128 + i] = acc2;
}
static uint16_t bf(float f){ uint32_t u; memcpy(&u,&f,4); return (uint16_t)(u>>16);} 
int main(){
    uint32_t ha[256], hb[256]; float ho[192];
    for(int i=0;i<256;i++){ float x0=1.0f+(i%7), x1=0.5f*(i%5)-3, y0=2.0f-(i%3), y1=-1.5f+(i%11)*0.25f; ha[i]=bf(x0)|((uint32_t)bf(x1)<<16); hb[i]=bf(y0)|((uint32_t)bf(y1)<<16);} 
    uint32_t *da,*db; float* dout; (void)hipMalloc(&da,1024); (void)hipMalloc(&db,1024); (void)hipMalloc(&dout,768);
    (void)hipMemcpy(da,ha,1024,hipMemcpyHostToDevice); (void)hipMemcpy(db,hb,1024,hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k,1,64,0,0,(const u32x4*)da,(const u32x4*)db,dout); (void)hipMemcpy(ho,dout,768,hipMemcpyDeviceToHost);
    int bad=0, bad2=0; for(int i=0;i<64;i++) { if (ho[i]!=ho[64+i]) bad++; if (ho[128+i]!=ho[64+i]) bad2++; }
    printf("inline-asm mismatches: %d of 64\n", bad2);
    for(int i=0;i<6;i++) printf("i=%d chained dot2=%g fma=%g\n", i, ho[i], ho[64+i]);
    printf("mismatches: %d of 64\n", bad);
    return 0; }
