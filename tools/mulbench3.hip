#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
// unsaturated 14 x 28-bit Montgomery product, product scanning
template<int L, int B>
__device__ __forceinline__ void umul(uint32_t (&r)[L], const uint32_t (&a)[L], const uint32_t (&b)[L], const uint32_t (&q)[L], uint32_t qinv) {
  constexpr uint32_t MASK = (1u<<B)-1;
  uint32_t m[L];
  uint64_t acc = 0;
#pragma unroll
  for (int k=0;k<L;k++){
#pragma unroll
    for (int i=0;i<=k;i++) acc += (uint64_t)a[i]*b[k-i];
#pragma unroll
    for (int i=0;i<k;i++) acc += (uint64_t)m[i]*q[k-i];
    m[k] = ((uint32_t)acc * qinv) & MASK;
    acc += (uint64_t)m[k]*q[0];
    acc >>= B;
  }
#pragma unroll
  for (int k=L;k<2*L-1;k++){
#pragma unroll
    for (int i=k-L+1;i<L;i++) acc += (uint64_t)a[i]*b[k-i];
#pragma unroll
    for (int i=k-L+1;i<L;i++) acc += (uint64_t)m[i]*q[k-i];
    r[k-L] = (uint32_t)acc & MASK;
    acc >>= B;
  }
  r[L-1] = (uint32_t)acc;
}
template<int L, int B>
__device__ __forceinline__ void umul2(uint32_t (&r)[L], const uint32_t (&a)[L], const uint32_t (&b)[L], const uint32_t (&q)[L], uint32_t qinv) {
  constexpr uint32_t MASK = (1u<<B)-1;
  uint32_t m[L];
  uint64_t carry = 0;
#pragma unroll
  for (int k=0;k<L;k++){
    uint64_t acc1 = carry, acc2 = 0;
#pragma unroll
    for (int i=0;i<=k;i++) acc1 += (uint64_t)a[i]*b[k-i];
#pragma unroll
    for (int i=0;i<k;i++) acc2 += (uint64_t)m[i]*q[k-i];
    uint64_t acc = acc1 + acc2;
    m[k] = ((uint32_t)acc * qinv) & MASK;
    acc += (uint64_t)m[k]*q[0];
    carry = acc >> B;
  }
#pragma unroll
  for (int k=L;k<2*L-1;k++){
    uint64_t acc1 = carry, acc2 = 0;
#pragma unroll
    for (int i=k-L+1;i<L;i++) acc1 += (uint64_t)a[i]*b[k-i];
#pragma unroll
    for (int i=k-L+1;i<L;i++) acc2 += (uint64_t)m[i]*q[k-i];
    uint64_t acc = acc1 + acc2;
    r[k-L] = (uint32_t)acc & MASK;
    carry = acc >> B;
  }
  r[L-1] = (uint32_t)carry;
}

// separated operand scanning: 2L independent 64-bit column accumulators (instruction-level parallelism for a wave that has
// its SIMD to itself), then the reduction row by row
template<int L, int B>
__device__ __forceinline__ void umul_sos(uint32_t (&r)[L], const uint32_t (&a)[L], const uint32_t (&b)[L], const uint32_t (&q)[L], uint32_t qinv) {
  constexpr uint32_t MASK = (1u<<B)-1;
  uint64_t T[2*L];
#pragma unroll
  for (int k=0;k<2*L;k++) T[k]=0;
#pragma unroll
  for (int i=0;i<L;i++){
#pragma unroll
    for (int j=0;j<L;j++) T[i+j] += (uint64_t)a[i]*b[j];
  }
#pragma unroll
  for (int k=0;k<L;k++){
    uint32_t m = ((uint32_t)T[k] * qinv) & MASK;
#pragma unroll
    for (int i=0;i<L;i++) T[k+i] += (uint64_t)m*q[i];
    T[k+1] += T[k] >> B;
  }
#pragma unroll
  for (int k=L;k<2*L-1;k++){
    r[k-L] = (uint32_t)T[k] & MASK;
    T[k+1] += T[k] >> B;
  }
  r[L-1] = (uint32_t)T[2*L-1];
}
// product scanning with the a b terms of a column in two chains (even / odd i) and the m q terms in a third
template<int L, int B>
__device__ __forceinline__ void umul3(uint32_t (&r)[L], const uint32_t (&a)[L], const uint32_t (&b)[L], const uint32_t (&q)[L], uint32_t qinv) {
  constexpr uint32_t MASK = (1u<<B)-1;
  uint32_t m[L];
  uint64_t carry = 0;
#pragma unroll
  for (int k=0;k<2*L-1;k++){
    uint64_t acc1 = carry, acc2 = 0, acc3 = 0;
    const int lo = k < L ? 0 : k-L+1, hi = k < L ? k : L-1;
#pragma unroll
    for (int i=lo;i<=hi;i++) { if (i&1) acc2 += (uint64_t)a[i]*b[k-i]; else acc1 += (uint64_t)a[i]*b[k-i]; }
    if (k < L) {
#pragma unroll
      for (int i=0;i<k;i++) acc3 += (uint64_t)m[i]*q[k-i];
      uint64_t acc = acc1 + acc2 + acc3;
      m[k] = ((uint32_t)acc * qinv) & MASK;
      acc += (uint64_t)m[k]*q[0];
      carry = acc >> B;
    } else {
#pragma unroll
      for (int i=k-L+1;i<L;i++) acc3 += (uint64_t)m[i]*q[k-i];
      uint64_t acc = acc1 + acc2 + acc3;
      r[k-L] = (uint32_t)acc & MASK;
      carry = acc >> B;
    }
  }
  r[L-1] = (uint32_t)carry;
}
template<int N>
__device__ __forceinline__ void smul(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N], const uint32_t (&q)[N], uint32_t inv) {
  uint32_t t[N];
#pragma unroll
  for (int i=0;i<N;i++) t[i]=0;
#pragma unroll
  for (int i=0;i<N;i++){
    uint64_t A = (uint64_t)a[0]*b[i] + t[0];
    uint32_t m = (uint32_t)A * inv;
    uint64_t C = (uint64_t)m*q[0] + (uint32_t)A;
    A >>= 32; C >>= 32;
#pragma unroll
    for (int j=1;j<N;j++){
      A += (uint64_t)a[j]*b[i] + t[j];
      C += (uint64_t)m*q[j] + (uint32_t)A;
      t[j-1] = (uint32_t)C;
      A >>= 32; C >>= 32;
    }
    t[N-1] = (uint32_t)(A + C);
  }
#pragma unroll
  for (int i=0;i<N;i++) r[i]=t[i];
}
#define ITERS 200
template<int L, int B, int UNSAT>
__global__ void k(uint32_t* out, const uint32_t* in, uint64_t* cyc){
  uint32_t a[L], b[L], q[L];
  int tid = blockIdx.x*blockDim.x+threadIdx.x;
  for (int i=0;i<L;i++){ a[i]=in[(tid*L+i)&1023] & ((1u<<28)-1); b[i]=in[(tid*L+i+7)&1023]& ((1u<<28)-1); q[i]=in[i+100] | 1; }
  uint64_t t0,t1;
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it=0; it<ITERS; ++it) { if constexpr (UNSAT==1) umul<L,B>(a,a,b,q,0x0ffcfffd); else if constexpr (UNSAT==2) umul2<L,B>(a,a,b,q,0x0ffcfffd); else if constexpr (UNSAT==3) umul_sos<L,B>(a,a,b,q,0x0ffcfffd); else if constexpr (UNSAT==4) umul3<L,B>(a,a,b,q,0x0ffcfffd); else smul<L>(a,a,b,q,0xfffcfffd); }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  for (int i=0;i<L;i++) out[tid*L+i]=a[i];
  if ((threadIdx.x&63)==0) cyc[tid>>6]=t1-t0;
}
template<int L,int B,int U> void run(const char* name){
  uint32_t *din,*dout; uint64_t* dc;
  hipMalloc(&din,4096*4); hipMalloc(&dout,256*2048*L*4); hipMalloc(&dc, 256*32*8);
  uint32_t h[1024]; for(int i=0;i<1024;i++) h[i]=i*2654435761u+12345; hipMemcpy(din,h,4096,hipMemcpyHostToDevice);
  for (int wps : {1,2,3,4}) {
    int threads=256*wps; 
    hipLaunchKernelGGL((k<L,B,U>), dim3(256), dim3(threads), 0, 0, dout, din, dc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<L,B,U>), dim3(256), dim3(threads), 0, 0, dout, din, dc);
    hipDeviceSynchronize();
    uint64_t c[1024]; hipMemcpy(c,dc,8*256*wps*4/ (wps*4) * 1,hipMemcpyDeviceToHost);
    printf("%s wps=%d cycles per wave-mul per SIMD = %.1f\n", name, wps, (double)c[0]/ITERS/wps);
  }
}
int main(){ run<14,29,1>("unsat14      "); run<14,29,2>("unsat14 split"); run<14,29,3>("unsat14 sos  "); run<14,29,4>("unsat14 3chn "); run<9,29,1>("unsat9       "); run<9,29,2>("unsat9 split "); return 0; }
