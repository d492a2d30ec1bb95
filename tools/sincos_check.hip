// Development tool (round 3): sincos_joint (qc_device.hpp) against the host's libm - random angles up to +-1e6 and +-1e9, +-10, +-3.25, and
// angles next to multiples of pi/2; NaN / Inf / -0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on tools/sincos_check.hip -o tools/_build/sincos_check
#include "../quadruped_control_amd/csrc/qc_device.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace qc;
__global__ void k(const double* x, double* s, double* c, int n) { int i = blockIdx.x*64+threadIdx.x; if (i<n) sincos_joint(x[i], &s[i], &c[i]); }
int main(){ const int n=1<<20; std::vector<double> h(n), hs(n), hc(n); for(int i=0;i<n;i++){ double u=(double)rand()/RAND_MAX; h[i]= (i%8==0)? (u-0.5)*2e9 : (i%4==0)? (u-0.5)*2e6 : (i%4==1? (u-0.5)*20 : (i%4==2? (round((u-0.5)*2000)*M_PI/2 + (u-0.5)*1e-9) : (u-0.5)*6.5)); }
 h[0]=0; h[1]=-0.0; h[2]=NAN; h[3]=INFINITY; h[4]=1073741823.5; h[7]=1073741824.0; h[8]=-3e9; h[5]=M_PI/4; h[6]=-M_PI/4;
 double *dx,*ds,*dc; hipMalloc(&dx,n*8); hipMalloc(&ds,n*8); hipMalloc(&dc,n*8); hipMemcpy(dx,h.data(),n*8,hipMemcpyHostToDevice);
 k<<<n/64,64>>>(dx,ds,dc,n); hipMemcpy(hs.data(),ds,n*8,hipMemcpyDeviceToHost); hipMemcpy(hc.data(),dc,n*8,hipMemcpyDeviceToHost);
 double ws=0,wc=0; int bad=0; for(int i=0;i<n;i++){ double rs=sin(h[i]), rc=cos(h[i]); if (std::isnan(rs) || fabs(h[i]) >= 1073741824.0) { if(!std::isnan(hs[i])||!std::isnan(hc[i])) bad++; continue;} ws=fmax(ws,fabs(hs[i]-rs)); wc=fmax(wc,fabs(hc[i]-rc)); }
 printf("max abs err sin %.3e cos %.3e, nan mismatches %d; sin(-0)=%g\n", ws, wc, bad, hs[1]); return 0; }
