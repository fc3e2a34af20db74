// CPU check of mh::sym_eigen3 (non-iterative, verified, Jacobi fallback) against mh::sym_eigen3_jacobi on 400 000 random symmetric
// PSD matrices: well separated, two eigenvalues within 1e-6 / 1e-9, rank deficient, multiples of the identity, scales 1e-3 .. 1e9.
#include <cstdio>
#include <random>
#include <cmath>
#include "../../mimosa_amd/csrc/math3.hpp"
int main() {
  std::mt19937_64 rng(3); std::normal_distribution<double> N(0,1); std::uniform_real_distribution<double> U(0,1);
  double worst_w=0, worst_v=0, worst_f=0, worst_o=0; long fast_ok=0, cases=0;
  for (int rep=0; rep<400000; ++rep) {
    // random PSD with controlled spectrum
    double Q[9]; { double a=N(rng),b=N(rng),c=N(rng),d=N(rng); double n=std::sqrt(a*a+b*b+c*c+d*d); a/=n;b/=n;c/=n;d/=n;
      Q[0]=1-2*(c*c+d*d);Q[1]=2*(b*c-a*d);Q[2]=2*(b*d+a*c);Q[3]=2*(b*c+a*d);Q[4]=1-2*(b*b+d*d);Q[5]=2*(c*d-a*b);Q[6]=2*(b*d-a*c);Q[7]=2*(c*d+a*b);Q[8]=1-2*(b*b+c*c);}
    double ev[3]; int kind=rep%6; double s=std::pow(10.0, 12*U(rng)-3);
    ev[0]=U(rng); ev[1]=ev[0]+U(rng); ev[2]=ev[1]+U(rng);
    if(kind==1){ev[1]=ev[0]*(1+1e-6*U(rng));} if(kind==2){ev[2]=ev[1]*(1+1e-9*U(rng));} if(kind==3){ev[0]=0;} if(kind==4){ev[0]=ev[1]=ev[2];} if(kind==5){ev[0]=1e-12*U(rng);ev[1]=1e-6*U(rng);}
    for(double&e:ev)e*=s;
    double A[9]={0}; for(int i=0;i<3;++i)for(int j=0;j<3;++j){double v=0;for(int k=0;k<3;++k)v+=Q[3*i+k]*ev[k]*Q[3*j+k];A[3*i+j]=v;}
    for(int i=0;i<3;++i)for(int j=i+1;j<3;++j)A[3*j+i]=A[3*i+j];
    double w[3],V[9],wj[3],Vj[9]; mh::sym_eigen3(A,w,V); mh::sym_eigen3_jacobi(A,wj,Vj);
    double sc=std::fabs(wj[2])+1e-300;
    for(int k=0;k<3;++k){ worst_w=std::fmax(worst_w,std::fabs(w[k]-wj[k])/sc);
      // residual of (w,V)
      double r=0; for(int i=0;i<3;++i){double av=0;for(int j=0;j<3;++j)av+=A[3*i+j]*V[3*j+k]; r=std::fmax(r,std::fabs(av-w[k]*V[3*i+k]));}
      worst_v=std::fmax(worst_v,r/sc);}
    ++cases;
    // the eigenvector-only solver of K4 (sym_eigvec3_fast): whatever it ACCEPTS must be the ascending eigenvectors, orthonormal
    double Vf[9];
    if (mh::sym_eigvec3_fast(A, Vf)) {
      ++fast_ok;
      for (int k = 0; k < 3; ++k) {
        double r = 0; for (int i = 0; i < 3; ++i) { double av = 0; for (int j = 0; j < 3; ++j) av += A[3*i+j] * Vf[3*j+k]; r = std::fmax(r, std::fabs(av - wj[k] * Vf[3*i+k])); }
        worst_f = std::fmax(worst_f, r / sc);
      }
      for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) { double d = 0; for (int i = 0; i < 3; ++i) d += Vf[3*i+a] * Vf[3*i+b]; worst_o = std::fmax(worst_o, std::fabs(d - (a == b))); }
    }
  }
  std::printf("cases %ld worst |dw|/|A| %.3e worst residual/|A| %.3e; fast vectors accepted %ld, worst residual %.3e, worst |V^T V - I| %.3e\n",cases,worst_w,worst_v,fast_ok,worst_f,worst_o);
  return (worst_w<1e-12&&worst_v<1e-12&&worst_f<1e-11&&worst_o<1e-9&&fast_ok>cases/10)?0:1;
}
