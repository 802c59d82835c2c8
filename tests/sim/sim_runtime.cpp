/* Workgroup executor of the SIMT mock (see include/hip/hip_runtime.h).  Test tool only. */
#include <hip/hip_runtime.h>

namespace sim {
Block *cur = nullptr;
unsigned grid_y = 0;
std::mutex &launch_mutex() { static std::mutex m; return m; }
thread_local unsigned tid = 0;

struct Job {
    unsigned grid, block, id;
    const std::function<void()> *fn;
    Block *blk;
};

static void *worker(void *p)
{
    Job *j = (Job *)p;
    tid = j->id;
    threadIdx = dim3(j->id, 0, 0);
    for (unsigned b = 0; b < j->grid; b++) {
        blockIdx = dim3(b, grid_y, 0);
        (*j->fn)();
        pthread_barrier_wait(&j->blk->bar); /* workgroups run one at a time: static LDS is reused */
    }
    return nullptr;
}

void launch(unsigned grid, unsigned block, const std::function<void()> &fn)
{
    if (grid == 0 || block == 0) return;
    if (block > 1024) { fprintf(stderr, "sim: block too large\n"); abort(); }
    Block blk;
    blk.nthreads = block;
    pthread_barrier_init(&blk.bar, nullptr, block);
    unsigned nw = (block + 63) / 64;
    for (unsigned w = 0; w < nw; w++) {
        unsigned cnt = block - 64 * w < 64 ? block - 64 * w : 64;
        pthread_barrier_init(&blk.wbar[w], nullptr, cnt);
    }
    cur = &blk;
    blockDim = dim3(block, 1, 1);
    gridDim = dim3(grid, 1, 1);
    std::vector<pthread_t> th(block);
    std::vector<Job> jobs(block);
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, 1 << 20);
    for (unsigned i = 0; i < block; i++) {
        jobs[i] = Job{grid, block, i, &fn, &blk};
        if (pthread_create(&th[i], &at, worker, &jobs[i]) != 0) { fprintf(stderr, "sim: pthread_create failed\n"); abort(); }
    }
    for (unsigned i = 0; i < block; i++) pthread_join(th[i], nullptr);
    pthread_attr_destroy(&at);
    pthread_barrier_destroy(&blk.bar);
    for (unsigned w = 0; w < nw; w++) pthread_barrier_destroy(&blk.wbar[w]);
    cur = nullptr;
}
}  // namespace sim

thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
