// Host-only check of the sp_lanc_eigh stopping rule (csrc/tridiag_host.hpp): the cheap pivot-based estimate never changes the iteration at
// which the rule fires -- the QL values decide, the estimate only skips QL sweeps that cannot fire.  Built and run by tests/test_stoprule_cpu.py.
#include <cstdio>
#include <random>
#include "../../dmft-lanc-ed_amd/csrc/tridiag_host.hpp"
using namespace edgpu;
int main(){
  std::mt19937 rng(5); std::uniform_real_distribution<double> U(-1,1);
  int bad=0; double maxd=0; long evals=0;
  for(int trial=0;trial<200;trial++){
    int n=5+trial%90; std::vector<double> al(n),be(n+1,0.0);
    for(int i=0;i<n;i++){al[i]=-5+3*U(rng)+ (trial%3==0? -0.1*i:0); be[i]= (i? 0.5+std::fabs(U(rng))*2:0.0);} 
    if(trial%7==0) be[n/2]=1e-9;   // nearly decoupled
    StopRule rule; double e0; int stopA=-1;
    for(int k=1;k<=n;k++){ int r=rule.check(al.data(),be.data(),k,10,1e-12,&e0); if(r>0){stopA=k;break;} }
    // reference rule
    double esave=0,ee; int stopB=-1;
    for(int k=1;k<=n;k++){ tridiag_lowest(al.data(),be.data(),k,&ee,nullptr); if(k>=10&&std::fabs(ee-esave)<=1e-12){stopB=k;break;} esave=ee; }
    if(stopA!=stopB){bad++; printf("trial %d n %d stopA %d stopB %d\n",trial,n,stopA,stopB);}
    for(int k=1;k<=n;k++){ double err; double est=tridiag_lowest_estimate(al.data(),be.data(),k,0,false,4e-15,&err); tridiag_lowest(al.data(),be.data(),k,&ee,nullptr); maxd=std::max(maxd,std::fabs(est-ee)/std::max(1.0,std::fabs(ee))); }
  }
  printf("mismatches %d  max rel diff estimate vs QL %.3e\n",bad,maxd);

  // Lanczos-generated T: converging lowest Ritz value
  int bad2=0, tot2=0;
  for(int seed=0; seed<60; seed++){
    std::mt19937 r2(100+seed); std::normal_distribution<double> N(0,1);
    int nn=400; std::vector<double> d(nn); for(int i=0;i<nn;i++) d[i]=-10+20.0*i/nn + (i==0? -(0.2+0.1*(seed%5)):0) + 0.01*N(r2);   // diagonal operator + tridiagonal coupling
    std::vector<double> off(nn,0.3+0.05*(seed%4));
    std::vector<double> v(nn),vp(nn,0.0),w(nn); double nr=0; for(auto&x:v){x=N(r2);nr+=x*x;} nr=std::sqrt(nr); for(auto&x:v)x/=nr;
    int K=150; std::vector<double> al(K),be(K+1,0.0); double beta=0;
    for(int k=0;k<K;k++){ for(int i=0;i<nn;i++){ w[i]=d[i]*v[i]+(i?off[i]*v[i-1]:0)+(i+1<nn?off[i+1]*v[i+1]:0)-beta*vp[i]; }
      double a=0; for(int i=0;i<nn;i++)a+=v[i]*w[i]; al[k]=a; double b2=0; for(int i=0;i<nn;i++){w[i]-=a*v[i]; b2+=w[i]*w[i];} beta=std::sqrt(b2); be[k+1]=beta; for(int i=0;i<nn;i++){vp[i]=v[i]; v[i]=w[i]/beta;} }
    for(double thr: {1e-10,1e-12,1e-13,1e-14}){
      StopRule rule; double e0,ee,esave=0; int stopA=-1, stopB=-1;
      for(int k=1;k<=K;k++){ int r=rule.check(al.data(),be.data(),k,10,thr,&e0); if(r>0){stopA=k;break;} }
      for(int k=1;k<=K;k++){ tridiag_lowest(al.data(),be.data(),k,&ee,nullptr); if(k>=10&&std::fabs(ee-esave)<=thr){stopB=k;break;} esave=ee; }
      tot2++; if(stopA!=stopB){bad2++; printf("lanczos seed %d thr %g stopA %d stopB %d\n",seed,thr,stopA,stopB);} else if(seed<3) printf("seed %d thr %g stop %d\n",seed,thr,stopA);
    }
  }
  printf("Lanczos-generated: %d mismatches of %d\n",bad2,tot2);
  return (bad || bad2) ? 1 : 0;
}
