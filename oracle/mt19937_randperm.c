/*
 * ORACLE -- test infrastructure only (never imported by the product path).
 *
 * CPU restatement of the random-index stream the reference consumes through
 * torch.randperm on the default CPU generator:
 *   - /root/reference/lib/loss/loss_contrast.py:79,81  (hard / easy draws)
 *   - /root/reference/lib/loss/loss_contrast_mem.py:79,81
 *   - /root/reference/segmentor/trainer_contrastive.py:127 (pixel-queue draw)
 *
 * Third-party arithmetic being restated (not under /root/reference): PyTorch
 * (requirements.txt:16 pins torch>=1.7.0; this container has 2.10.0).
 *   torch.manual_seed(s)  == mt19937 init_genrand((uint32)s)   (Matsumoto & Nishimura 1998/2002)
 *   torch.randperm(n)     == forward Fisher-Yates, for i in [0, n-1): z = genrand_int32() % (n - i);
 *                            swap(r[i], r[i+z])   -> consumes max(n-1, 0) 32-bit draws.
 * Pinned in tests/test_oracle_rng.py against torch.randperm itself.
 */
#include <stdint.h>
#include <stddef.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t mt[MT_N];
    int idx;
} mt19937_t;

void mt_seed(mt19937_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < MT_N; ++i)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = MT_N;
}

static void mt_refill(mt19937_t *s) {
    uint32_t *mt = s->mt;
    for (int k = 0; k < MT_N; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % MT_N] & 0x7fffffffu);
        uint32_t v = mt[(k + MT_M) % MT_N] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[k] = v;
    }
    s->idx = 0;
}

uint32_t mt_next(mt19937_t *s) {
    if (s->idx >= MT_N) mt_refill(s);
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* out[0..n) = the permutation torch.randperm(n) returns from this generator state. */
void mt_randperm(mt19937_t *s, int64_t n, int64_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    for (int64_t i = 0; i + 1 < n; ++i) {
        int64_t z = (int64_t)(mt_next(s) % (uint32_t)(n - i));
        int64_t t = out[i]; out[i] = out[i + z]; out[i + z] = t;
    }
}

size_t mt_state_size(void) { return sizeof(mt19937_t); }
