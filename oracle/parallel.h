// ORACLE (test infrastructure only).  Row / point parallelism of the CPU restatement: every parallel loop computes independent outputs,
// so the results do not depend on the thread count.  lvko_set_num_threads() fixes the count for the stages that take no `nthreads`
// argument of their own (tracking-frame downscale, optical flow, 4:2:0 conversion); the stabilizer's push sets it from its argument.
#pragma once
#include <algorithm>
#include <thread>
#include <vector>

extern "C" int lvko_set_num_threads(int n);      // returns the previous value; n < 1 = 1
int lvko_num_threads();

template <class Fn>
inline void lvko_parallel_for(int count, int min_chunk, Fn fn)      // fn(begin, end)
{
    const int nt = std::min(lvko_num_threads(), std::max(1, count / std::max(1, min_chunk)));
    if (nt <= 1) { fn(0, count); return; }
    std::vector<std::thread> pool;
    const int chunk = (count + nt - 1) / nt;
    for (int t = 0; t < nt; t++)
    {
        const int b = t * chunk, e = std::min(count, b + chunk);
        if (b >= e) break;
        pool.emplace_back([=] { fn(b, e); });
    }
    for (auto& th : pool) th.join();
}
