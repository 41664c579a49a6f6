/*
 * ORACLE — TEST INFRASTRUCTURE ONLY, build container only, DIAGNOSTIC ONLY (not a parity pin).
 *
 * A CPU execution model for the TEXT of the reference's CUDA kernels (sampling_gpu.cu, ball_query_gpu.cu, interpolate_gpu.cu).
 * tests/test_cu_text_emulation.py extracts the kernel bodies from /root/reference at test time into a temporary directory
 * (nothing of the reference is committed or shipped), compiles them with the host compilers against this header and runs them:
 *   - every CUDA thread of a block is a cooperative fiber; __syncthreads() yields to a round-robin scheduler that resumes the
 *     block's fibers in threadIdx order until all of them have arrived (or returned); blocks run one after another;
 *   - __shared__ is `static` (one block at a time), threadIdx / blockIdx / blockDim / gridDim are globals set by the scheduler;
 *   - max / min on floats are fmaxf / fminf (CUDA's overloads).
 * What this can and cannot show: it executes the kernel text itself, so "the C restatement follows the .cu" becomes a test; and the
 * HOST compiler decides how `dx*dx + dy*dy + dz*dz` contracts (-ffp-contract=off / fast), which is evidence for - not proof of -
 * what nvcc's LLVM-derived compiler emits.  It stands in for the CUDA headers, so it is NOT an oracle/_ref build and pins nothing.
 */
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <functional>
#include <vector>

struct emu_dim3 { unsigned x, y, z; };
static emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __shared__ static
#define __restrict__

static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }

/* ---- fibers: a 7-register context switch (System V x86-64: rbx rbp r12-r15 + rsp) */
extern "C" void emu_switch(void **save_sp, void *load_sp);
__asm__(
    ".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_switch,.-emu_switch\n");

struct EmuFiber { void *sp; char *stack; bool done; };
static std::vector<EmuFiber> emu_fibers;
static void *emu_sched_sp;
static int emu_cur;
static const std::function<void()> *emu_body;

static inline void __syncthreads() { emu_switch(&emu_fibers[emu_cur].sp, emu_sched_sp); }

static void emu_entry() {
    (*emu_body)();
    emu_fibers[emu_cur].done = true;
    emu_switch(&emu_fibers[emu_cur].sp, emu_sched_sp);
    abort(); /* a finished fiber is never resumed */
}

/* run one block of `nthreads` (x dimension) threads through `body` */
static void emu_run_block(unsigned bx, unsigned by, unsigned bz, unsigned nthreads, const std::function<void()> &body) {
    const size_t STACK = 64 * 1024;
    blockIdx = {bx, by, bz};
    blockDim = {nthreads, 1, 1};
    emu_body = &body;
    emu_fibers.assign(nthreads, EmuFiber{nullptr, nullptr, false});
    for (unsigned t = 0; t < nthreads; ++t) {
        char *st = (char *)aligned_alloc(64, STACK);
        emu_fibers[t].stack = st;
        uintptr_t top = ((uintptr_t)(st + STACK)) & ~(uintptr_t)15;
        void **p = (void **)top;
        *--p = nullptr;             /* return address of emu_entry (never used) */
        *--p = (void *)&emu_entry;  /* `ret` of the first switch jumps here: rsp = top - 8 at entry, as the ABI wants */
        for (int r = 0; r < 6; ++r) *--p = nullptr;
        emu_fibers[t].sp = (void *)p;
    }
    unsigned alive = nthreads;
    while (alive) {
        for (unsigned t = 0; t < nthreads; ++t) {
            if (emu_fibers[t].done) continue;
            emu_cur = (int)t;
            threadIdx = {t, 0, 0};
            emu_switch(&emu_sched_sp, emu_fibers[t].sp);
            if (emu_fibers[t].done) --alive;
        }
    }
    for (unsigned t = 0; t < nthreads; ++t) free(emu_fibers[t].stack);
}
