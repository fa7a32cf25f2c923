// Experiment: which part of vox_dense_kernel costs time?  MODE bits:
//  1: load cnt + scan + barrier   2: gather to LDS   4: compute from LDS (else zeros)  8: skip P==0 fast path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
constexpr int CHUNK=1024, VAL_FLOATS=8192;
__device__ __forceinline__ int wave_incl_scan(int v,int lane){ for(int d=1;d<64;d<<=1){int t=__shfl_up(v,d,64); if(lane>=d) v+=t;} return v;}
template<int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ feat,const int* __restrict__ cnt,const int* __restrict__ sorted,const int* __restrict__ chunk_start,int C,int N,int r3,int nchunks,int CT,float* __restrict__ out){
  __shared__ float vals[VAL_FLOATS]; __shared__ int wt[4];
  const int tid=threadIdx.x,lane=tid&63,wave=tid>>6; const int chunk=blockIdx.x,b=blockIdx.z;
  const int c_begin=blockIdx.y*CT,c_end=min(C,c_begin+CT); const int v0=chunk*CHUNK+tid*4; const bool inb=v0<r3;
  float* obase=out+((size_t)b*C)*r3+v0;
  int P=0,cs0=0;
  if(MODE&1){ cs0=chunk_start[(size_t)b*(nchunks+1)+chunk]; P=chunk_start[(size_t)b*(nchunks+1)+chunk+1]-cs0; }
  if(P==0){ if(inb){ const float4 z=make_float4(0,0,0,0);
#pragma unroll 8
    for(int c=c_begin;c<c_end;++c) *reinterpret_cast<float4*>(obase+(size_t)c*r3)=z;} return; }
  int c4[4]={0,0,0,0};
  if(inb){ const int4 t=*reinterpret_cast<const int4*>(cnt+(size_t)b*r3+v0); c4[0]=t.x;c4[1]=t.y;c4[2]=t.z;c4[3]=t.w; }
  const int total=c4[0]+c4[1]+c4[2]+c4[3]; const int incl=wave_incl_scan(total,lane); if(lane==63) wt[wave]=incl;
  const int* srt=sorted+(size_t)b*N+cs0; const int cpp=max(1,min(c_end-c_begin,VAL_FLOATS/P));
  if(MODE&2){ const int nc=min(cpp,c_end-c_begin); const float* fbase=feat+((size_t)b*C+c_begin)*N;
    for(int e=tid;e<nc*P;e+=256){ const int cc=e/P,j=e-cc*P; vals[e]=fbase[(size_t)cc*N+srt[j]]; } }
  __syncthreads();
  int rel=incl-total; for(int w=0;w<wave;++w) rel+=wt[w];
  float inv[4]; int off[4];
#pragma unroll
  for(int q=0;q<4;++q){ inv[q]=c4[q]>0?1.0f/(float)c4[q]:0.f; off[q]=(q==0)?rel:off[q-1]+c4[q-1]; }
  const int nc=min(cpp,c_end-c_begin);
  for(int cc=0;cc<nc;++cc){ float val[4]={0,0,0,0};
    if((MODE&4) && total>0){ const float* vrow=vals+cc*P;
#pragma unroll
      for(int q=0;q<4;++q){ float acc=0; for(int kk=0;kk<c4[q];++kk) acc=acc+vrow[off[q]+kk]*inv[q]; val[q]=acc; } }
    if(inb) *reinterpret_cast<float4*>(obase+(size_t)(c_begin+cc)*r3)=make_float4(val[0],val[1],val[2],val[3]); }
}
int main(){
  const int B=32,C=64,N=2048,r=32,r3=r*r*r,nchunks=r3/CHUNK,CT=16;
  std::mt19937 g(0); std::normal_distribution<float> nd(0,1);
  std::vector<int> cnt((size_t)B*r3,0), sorted((size_t)B*N), cs((size_t)B*(nchunks+1)); std::vector<float> feat((size_t)B*C*N,1.f);
  for(int b=0;b<B;++b){ std::vector<float> p(3*N); float m[3]={0,0,0}; for(int i=0;i<3*N;++i){p[i]=nd(g); m[i/N]+=p[i]/N;} float mx=0; for(int i=0;i<N;++i){float x=p[i]-m[0],y=p[N+i]-m[1],z=p[2*N+i]-m[2]; mx=std::max(mx,std::sqrt(x*x+y*y+z*z));}
    std::vector<std::pair<int,int>> vi(N); for(int i=0;i<N;++i){ int q[3]; for(int a=0;a<3;++a){ float v=(p[a*N+i]-m[a])/(2*mx)+0.5f; v=std::min(std::max(v*r,0.f),(float)(r-1)); q[a]=(int)std::nearbyint(v);} int v=q[0]*r*r+q[1]*r+q[2]; vi[i]={v,i}; cnt[(size_t)b*r3+v]++; }
    std::sort(vi.begin(),vi.end()); for(int i=0;i<N;++i) sorted[(size_t)b*N+i]=vi[i].second;
    int run=0; for(int q=0;q<=nchunks;++q){ cs[(size_t)b*(nchunks+1)+q]=run; if(q<nchunks) for(int v=q*CHUNK;v<(q+1)*CHUNK;++v) run+=cnt[(size_t)b*r3+v]; } }
  int nonempty=0; for(int b=0;b<B;++b) for(int q=0;q<nchunks;++q) nonempty+= cs[(size_t)b*(nchunks+1)+q+1]>cs[(size_t)b*(nchunks+1)+q]; printf("non-empty chunks: %d of %d\n",nonempty,B*nchunks);
  float *dfeat,*dout; int *dcnt,*dsorted,*dcs; hipMalloc(&dfeat,feat.size()*4); hipMalloc(&dout,(size_t)B*C*r3*4); hipMalloc(&dcnt,cnt.size()*4); hipMalloc(&dsorted,sorted.size()*4); hipMalloc(&dcs,cs.size()*4);
  hipMemcpy(dfeat,feat.data(),feat.size()*4,hipMemcpyHostToDevice); hipMemcpy(dcnt,cnt.data(),cnt.size()*4,hipMemcpyHostToDevice); hipMemcpy(dsorted,sorted.data(),sorted.size()*4,hipMemcpyHostToDevice); hipMemcpy(dcs,cs.data(),cs.size()*4,hipMemcpyHostToDevice);
  hipEvent_t a,e; hipEventCreate(&a); hipEventCreate(&e); float ms; const int IT=20; dim3 grid(nchunks,C/CT,B);
#define RUN(M) { for(int w=0;w<2;++w){ hipEventRecord(a); for(int i=0;i<IT;++i) k<M><<<grid,256>>>(dfeat,dcnt,dsorted,dcs,C,N,r3,nchunks,CT,dout); hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms,a,e);} printf("MODE %2d: %7.1f us  %7.1f GB/s\n",M,ms/IT*1e3,(double)B*C*r3*4/(ms/IT*1e-3)/1e9); }
  RUN(0) RUN(1) RUN(3) RUN(5) RUN(7)
  printf("%s\n",hipGetErrorString(hipGetLastError()));
  return 0; }
