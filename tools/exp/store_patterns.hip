// Experiment: what write bandwidth can a [B,C,r3] dense zero-ish fill reach on MI355X, by pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ __launch_bounds__(256) void linear_fill(float4* o, size_t n4){
  size_t i = (size_t)blockIdx.x*256+threadIdx.x; size_t stride=(size_t)gridDim.x*256;
  for(; i<n4; i+=stride) o[i]=make_float4(0,0,0,0);
}
// WG = chunk of 1024 voxels x CT channels (stride r3 between channels)
template<int NT>
__global__ __launch_bounds__(256) void chunk_fill(float* out, int C, int r3, int CT){
  const int chunk=blockIdx.x, b=blockIdx.z; const int c0=blockIdx.y*CT, c1=min(C,c0+CT);
  float* ob = out + ((size_t)b*C)*r3 + chunk*1024 + threadIdx.x*4;
  const float4 z=make_float4(0,0,0,0);
#pragma unroll 8
  for(int c=c0;c<c1;++c){
    if(NT){ v4f zz={0,0,0,0}; __builtin_nontemporal_store(zz, reinterpret_cast<v4f*>(ob+(size_t)c*r3)); }
    else *reinterpret_cast<float4*>(ob+(size_t)c*r3)=z;
  }
}
// WG = (b, c) whole slab contiguous: 256 threads x float4 x loop
__global__ __launch_bounds__(256) void slab_fill(float* out, int r3, int slabs_per_wg){
  size_t base = (size_t)blockIdx.x*slabs_per_wg*r3;
  float4* o=reinterpret_cast<float4*>(out+base); size_t n4=(size_t)slabs_per_wg*r3/4;
  const float4 z=make_float4(0,0,0,0);
  for(size_t i=threadIdx.x;i<n4;i+=256) o[i]=z;
}
int main(){
  const int B=32,C=64,r3=32768; size_t n=(size_t)B*C*r3; float* d; CK(hipMalloc(&d,n*4));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); float ms;
  auto rep=[&](const char* name, float ms_, int it){ printf("%-40s %8.1f us  %7.1f GB/s\n", name, ms_/it*1e3, n*4.0/(ms_/it*1e-3)/1e9); };
  const int IT=20;
  for(int w=0;w<2;++w){
  hipEventRecord(a); for(int i=0;i<IT;++i) hipMemsetAsync(d,0,n*4); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); rep("hipMemsetAsync",ms,IT);
  for(int g: {1024,2048,4096,8192,16384}){ hipEventRecord(a); for(int i=0;i<IT;++i) linear_fill<<<g,256>>>((float4*)d,n/4); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); char nm[64]; sprintf(nm,"linear_fill grid=%d",g); rep(nm,ms,IT);}
  for(int ct: {4,8,16,32,64}){ dim3 g(r3/1024,(C+ct-1)/ct,B); hipEventRecord(a); for(int i=0;i<IT;++i) chunk_fill<0><<<g,256>>>(d,C,r3,ct); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); char nm[64]; sprintf(nm,"chunk_fill CT=%d",ct); rep(nm,ms,IT);
    hipEventRecord(a); for(int i=0;i<IT;++i) chunk_fill<1><<<g,256>>>(d,C,r3,ct); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); sprintf(nm,"chunk_fill NT CT=%d",ct); rep(nm,ms,IT);}
  for(int s: {1,2,4}){ hipEventRecord(a); for(int i=0;i<IT;++i) slab_fill<<<B*C/s,256>>>(d,r3,s); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); char nm[64]; sprintf(nm,"slab_fill slabs/wg=%d",s); rep(nm,ms,IT);}
  }
  CK(hipGetLastError());
  return 0;
}
