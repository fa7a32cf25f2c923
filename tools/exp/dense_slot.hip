// Experiment: proposed K_dense = load int4 slot -> (occupied lanes) gather table[c][slot] -> float4 stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
template<int DEPTH,int CT, int NT>
__global__ __launch_bounds__(NT) void k(const int* __restrict__ vslot,const float* __restrict__ vmean,int C,int U,int r3,float* __restrict__ out){
  const int b=blockIdx.z; const int c0=blockIdx.y*CT; const int v0=(blockIdx.x*NT+threadIdx.x)*4; if(v0>=r3) return;
  float* ob=out+((size_t)b*C+c0)*r3+v0;
  int4 s=make_int4(-1,-1,-1,-1);
  if(DEPTH>=1) s=*reinterpret_cast<const int4*>(vslot+(size_t)b*r3+v0);
  const float* tb=vmean+((size_t)b*C+c0)*U;
  if(DEPTH<2 || (s.x&s.y&s.z&s.w)<0){ // all empty (all -1) or no gather
    float z = DEPTH==1 ? (float)(s.x+1) : 0.f;
#pragma unroll
    for(int c=0;c<CT;++c) *reinterpret_cast<float4*>(ob+(size_t)c*r3)=make_float4(z,0,0,0);
  } else {
    float4 v[CT];
#pragma unroll
    for(int c=0;c<CT;++c){ v[c].x = s.x>=0? tb[(size_t)c*U+s.x]:0.f; v[c].y = s.y>=0? tb[(size_t)c*U+s.y]:0.f; v[c].z = s.z>=0? tb[(size_t)c*U+s.z]:0.f; v[c].w = s.w>=0? tb[(size_t)c*U+s.w]:0.f; }
#pragma unroll
    for(int c=0;c<CT;++c) *reinterpret_cast<float4*>(ob+(size_t)c*r3)=v[c];
  }
}
int main(){
  const int B=32,C=64,N=2048,r=32,r3=r*r*r;
  std::mt19937 g(0); std::normal_distribution<float> nd(0,1);
  std::vector<int> slot((size_t)B*r3,-1); int Umax=0;
  for(int b=0;b<B;++b){ std::vector<float> p(3*N); float m[3]={0,0,0}; for(int i=0;i<3*N;++i){p[i]=nd(g); m[i/N]+=p[i]/N;} float mx=0; for(int i=0;i<N;++i){float x=p[i]-m[0],y=p[N+i]-m[1],z=p[2*N+i]-m[2]; mx=std::max(mx,std::sqrt(x*x+y*y+z*z));}
    std::vector<int> occ; for(int i=0;i<N;++i){ int q[3]; for(int a=0;a<3;++a){ float v=(p[a*N+i]-m[a])/(2*mx)+0.5f; v=std::min(std::max(v*r,0.f),(float)(r-1)); q[a]=(int)std::nearbyint(v);} occ.push_back(q[0]*r*r+q[1]*r+q[2]); }
    std::sort(occ.begin(),occ.end()); occ.erase(std::unique(occ.begin(),occ.end()),occ.end()); for(size_t u=0;u<occ.size();++u) slot[(size_t)b*r3+occ[u]]=u; Umax=std::max(Umax,(int)occ.size()); }
  printf("Umax=%d\n",Umax); const int U=N;
  float *dmean,*dout; int* dslot; hipMalloc(&dmean,(size_t)B*C*U*4); hipMemset(dmean,0,(size_t)B*C*U*4); hipMalloc(&dout,(size_t)B*C*r3*4); hipMalloc(&dslot,slot.size()*4); hipMemcpy(dslot,slot.data(),slot.size()*4,hipMemcpyHostToDevice);
  hipEvent_t a,e; hipEventCreate(&a); hipEventCreate(&e); float ms; const int IT=20;
#define RUN(D,CT,NT) { dim3 grid(r3/(NT*4),C/CT,B); for(int w=0;w<2;++w){ hipEventRecord(a); for(int i=0;i<IT;++i) k<D,CT,NT><<<grid,NT>>>(dslot,dmean,C,U,r3,dout); hipEventRecord(e); hipEventSynchronize(e); hipEventElapsedTime(&ms,a,e);} printf("DEPTH %d CT %2d NT %3d: %7.1f us  %7.1f GB/s\n",D,CT,NT,ms/IT*1e3,(double)B*C*r3*4/(ms/IT*1e-3)/1e9); }
  RUN(0,16,256) RUN(1,16,256) RUN(2,16,256) RUN(2,8,256) RUN(2,4,256) RUN(2,16,64) RUN(2,8,64) RUN(2,8,128) RUN(2,16,128) RUN(2,16,512) RUN(2,8,512)
  printf("%s\n",hipGetErrorString(hipGetLastError()));
  return 0; }
