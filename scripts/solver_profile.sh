# Section profile of csrc/lbfgsb.cpp on a route!-shaped problem (rdtsc ticks per section, instrumented COPY of the source built with clang).
# usage: bash scripts/solver_profile.sh [n_tokens] [reps]
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && cp $R/cfmmrouter.jl_amd/csrc/lbfgsb.cpp lb.cpp && cp $R/cfmmrouter.jl_amd/csrc/lbfgsb.h . && R=$R python - <<'EOF'
import os
s=open('/tmp/lb.cpp').read()
def rep(a,b):
    global s
    assert a in s, a[:60]
    s=s.replace(a,b,1)
rep('#include <numeric>','#include <numeric>\n#include <x86intrin.h>\n#include <cstdio>\n#include <chrono>\nstatic unsigned long long T_[24]; static double TSC_MHZ=2100.0; static unsigned long long tsc0_=__rdtsc(); static auto clk0_=std::chrono::steady_clock::now(); static unsigned long long t_last;\n#define TICK(k) do{unsigned long long n_=__rdtsc(); T_[k]+=n_-t_last; t_last=n_;}while(0)\nextern "C" void prof_dump(int runs,int iters){TSC_MHZ=(__rdtsc()-tsc0_)/std::chrono::duration<double,std::micro>(std::chrono::steady_clock::now()-clk0_).count();printf("TSC %.0f MHz\\n",TSC_MHZ);const char*nm[]={"other","cauchy_setup","cauchy_bp","subspace_tail","ls_after_eval","eval","update","x","cs_loop","cs_heap","cs_wt","sub_r","sub_wzr","sub_gram","sub_N","sub_du","ls_setup","ls_x","term"};double tot=0;for(int i=0;i<19;i++){if(i!=5)tot+=T_[i];printf("%-14s %7.2f us/iter\\n",nm[i],T_[i]/TSC_MHZ/runs/iters);}printf("total w/o eval %.2f us/iter\\n",tot/TSC_MHZ/runs/iters);}\n')
rep('    int iter = 0;\n    for (;;) {','    int iter = 0;\n    t_last=__rdtsc();\n    for (;;) {\n        TICK(0);')
rep('            auto later = [&](int a, int b)','            TICK(8);\n            auto later = [&](int a, int b)')
rep('            mem.Wt_times(d.data(), p.data());   // p = W\'d','            TICK(9);\n            mem.Wt_times(d.data(), p.data());   // p = W\'d')
rep('            const double fpp_org = fpp;','            TICK(1);\n            const double fpp_org = fpp;')
rep('        // ---------------- subspace minimization','        TICK(2);\n        // ---------------- subspace minimization')
rep('            wzr.assign(k2, 0.0);\n','            TICK(11);\n            wzr.assign(k2, 0.0);\n')
rep('            const int nfix = n - nf;','            TICK(12);\n            const int nfix = n - nf;')
rep('            std::vector<double>& v = vv;','            TICK(13);\n            std::vector<double>& v = vv;')
rep('            if (solve_dense(N, v, k2, 1)) {','            TICK(14);\n            if (solve_dense(N, v, k2, 1)) {')
rep('        // ---------------- line search along d = z − x','        TICK(3);\n        // ---------------- line search along d = z − x')
rep('        double stp = 1.0, f_old = f;','        TICK(15);\n        double stp = 1.0, f_old = f;')
rep('            while (!ls_failed && task == MoreThuente::kEvaluate) {','            TICK(16);\n            while (!ls_failed && task == MoreThuente::kEvaluate) {')
rep('                const double fnew = evaluate(x, g.data());','                TICK(17);\n                const double fnew = evaluate(x, g.data());\n                TICK(5);')
rep('        ++iter;\n','        TICK(4);\n        ++iter;\n')
rep('        // ---------------- limited-memory update ---','        TICK(18);\n        // ---------------- limited-memory update ---')
rep('        if (sy > kEps * (-sg_old)) mem.push(s, y, sy, yy); // else: curvature too small, skip','        if (sy > kEps * (-sg_old)) mem.push(s, y, sy, yy); // else: curvature too small, skip\n        TICK(6);')
open('/tmp/lb.cpp','w').write(s)
b=open(os.environ['R']+'/scripts/native/solver_bench.cpp').read()
b=b.replace('#include "cfmm_amd.h"','#include "lbfgsb.h"\nextern "C" void prof_dump(int,int);\nstruct cfmm_route_info{int iterations,status;};')
b=b.replace('        cfmm_lbfgsb_minimize(n, x.data(), lo.data(), hi.data(), nbd.data(), fg, &M, 5, 1e1, 1e-5, 15000, 15000, 1, &info);','        cfmm::LbfgsbOptions o; o.boxed_from_nbd=true; auto r_=cfmm::lbfgsb_minimize(n, x.data(), lo.data(), hi.data(), nbd.data(), [&](const double*xx,double*gg){return fg(&M,xx,gg);}, o); info.iterations=r_.iterations; info.status=r_.status;')
b=b.replace('std::vector<int32_t> nbd','std::vector<int> nbd')
b=b.replace('    return 0;\n}','    prof_dump(reps, info.iterations);\n    return 0;\n}')
open('/tmp/sb.cpp','w').write(b)
EOF
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -mavx2 -ffp-contract=off -o sb.bin sb.cpp lb.cpp && ./sb.bin ${1:-512} 20000 ${2:-40}
