// Checks div3 (fnn_core.h) against the IEEE division on random and structured operands: gcc -O2 -mfma -fopenmp -ffp-contract=off div3_check.c -lm
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <omp.h>
static inline double u2f(uint64_t u){double d;memcpy(&d,&u,8);return d;}
static inline uint64_t f2u(double d){uint64_t u;memcpy(&u,&d,8);return u;}
static inline double div3(double x){
    const double z = 0x1.5555555555555p-2; // RN(1/3)
    double q = x*z;
    double r = __builtin_fma(-3.0, q, x);
    return __builtin_fma(r, z, q);
}
static inline uint64_t sm(uint64_t* s){uint64_t z=(*s+=0x9E3779B97F4A7C15ULL);z=(z^(z>>30))*0xBF58476D1CE4E5B9ULL;z=(z^(z>>27))*0x94D049BB133111EBULL;return z^(z>>31);}
int main(){
    long bad=0; long total=0;
    #pragma omp parallel reduction(+:bad,total)
    {
        uint64_t s=1234567+omp_get_thread_num()*7919;
        for(long i=0;i<400000000L;i++){
            uint64_t r=sm(&s);
            // random mantissa, exponent in the safe range [-900, 900] (biased 123..1923), random sign
            uint64_t e = 123 + (sm(&s) % 1801);
            uint64_t u = (r & 0x800FFFFFFFFFFFFFULL) | (e<<52);
            double x=u2f(u);
            volatile double t = x/3.0;
            double d=div3(x);
            if(f2u(d)!=f2u(t)){ if(bad<5) printf("BAD x=%a div=%a fast=%a\n",x,(double)t,d); bad++; }
            total++;
        }
        // structured: multiples of 3 +- few ulps, mantissas near patterns
        for(long k=1;k<30000000L;k++){
            double b=(double)(3*k);
            for(int e=-3;e<=3;e++){ double x=u2f(f2u(b)+e); volatile double t=x/3.0; double d=div3(x); if(f2u(d)!=f2u(t)){ if(bad<5) printf("BAD2 x=%a\n",x); bad++;} total++; }
            // large mantissa values: k * 2^30-ish odd patterns
            double y=u2f(0x3FF0000000000000ULL + (uint64_t)k*0x9E3779B97F4AULL % (1ULL<<52));
            volatile double t2=y/3.0; double d2=div3(y); if(f2u(d2)!=f2u(t2)){ if(bad<5) printf("BAD3 y=%a\n",y); bad++;} total++;
        }
    }
    printf("total=%ld bad=%ld\n",total,bad);
    return bad!=0;
}
