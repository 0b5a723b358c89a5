/* selfcheck.c -- runs every exported function of glu_oracle.c on the reference tests' inputs under AddressSanitizer and
 * UndefinedBehaviorSanitizer (CPU build only; `make -C oracle sanitize`).  Test infrastructure, like the oracle itself:
 * the two restatements (literal reference algorithm, LSD checker) must agree with each other and with a plain qsort, and
 * must not touch a byte outside their arguments. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

uint64_t glu_oracle_div_ceil(uint64_t n, uint64_t d);
uint64_t glu_oracle_next_power_of_2(uint64_t n);
void glu_oracle_minstd_sample(uint64_t seed, uint64_t n, uint32_t min, uint32_t max, uint32_t* out);
int glu_oracle_blelloch_scan_u32(uint32_t* data, uint64_t count, uint64_t num_partitions);
void glu_oracle_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t count, uint64_t num_partitions);
uint64_t glu_oracle_radix_scratch_buffer_size(uint64_t count);
int glu_oracle_radix_sort_reference(uint32_t* key, uint32_t* val, uint32_t* key_scratch, uint32_t* val_scratch, uint64_t count,
                                    uint64_t num_steps, uint32_t* trace_block_count, int* result_in_scratch);
int glu_oracle_stable_sort_pairs_u32(uint32_t* key, uint32_t* val, uint64_t count, uint32_t key_bits);
int glu_oracle_stable_sort_pairs_u64(uint64_t* key, uint32_t* val, uint64_t count, uint32_t key_bits);
int glu_oracle_reduce_reference_u32(uint32_t* data, uint64_t count, int op, uint32_t subgroup_size);

static int failures = 0;
#define EXPECT(cond)                                                                 \
    do                                                                               \
    {                                                                                \
        if (!(cond))                                                                 \
        {                                                                            \
            printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                 \
            failures++;                                                              \
        }                                                                            \
    } while (0)

static void check_sort(uint64_t n, uint32_t key_max, uint64_t steps)
{
    /* exactly-sized heap blocks: any access past either end is an ASan report */
    uint32_t* key = malloc(n * 4 + (n == 0));
    uint32_t* val = malloc(n * 4 + (n == 0));
    uint32_t* ks = malloc(n * 4 + (n == 0));
    uint32_t* vs = malloc(n * 4 + (n == 0));
    uint32_t* k2 = malloc(n * 4 + (n == 0));
    uint32_t* v2 = malloc(n * 4 + (n == 0));
    glu_oracle_minstd_sample(1, n, 0, key_max, key);
    for (uint64_t i = 0; i < n; i++)
    {
        key[i] ^= (uint32_t) (i * 2654435761u) & 0x80000000u; /* the reference's generator never sets bit 31 */
        val[i] = (uint32_t) i;
    }
    memcpy(k2, key, n * 4);
    memcpy(v2, val, n * 4);
    int in_scratch = 0;
    int passes = glu_oracle_radix_sort_reference(key, val, ks, vs, n, steps, NULL, &in_scratch);
    const uint32_t* rk = in_scratch ? ks : key;
    const uint32_t* rv = in_scratch ? vs : val;
    uint32_t bits = n <= 1 ? 0 : (uint32_t) passes * 4;
    glu_oracle_stable_sort_pairs_u32(k2, v2, n, bits);
    EXPECT(n <= 1 || passes == (int) ((steps == 0 || steps > 8) ? 8 : steps));
    EXPECT(n <= 1 || (memcmp(rk, k2, n * 4) == 0 && memcmp(rv, v2, n * 4) == 0));
    free(key); free(val); free(ks); free(vs); free(k2); free(v2);
}

int main(void)
{
    const uint64_t sizes[] = {0, 1, 2, 128, 1023, 1024, 1025, 2048, 10993, 47487};
    for (unsigned i = 0; i < sizeof(sizes) / sizeof(sizes[0]); i++)
        for (uint64_t steps = 0; steps <= 9; steps += (sizes[i] > 4000 ? 4 : 1))
            check_sort(sizes[i], sizes[i] == 2048 ? 10u : 0xFFFFFFFFu, steps);

    /* 64-bit checker against the 32-bit one on zero-extended keys */
    {
        const uint64_t n = 5003;
        uint32_t* k = malloc(n * 4); uint32_t* v = malloc(n * 4); uint64_t* k64 = malloc(n * 8); uint32_t* v64 = malloc(n * 4);
        glu_oracle_minstd_sample(7, n, 0, 1000, k);
        for (uint64_t i = 0; i < n; i++) { v[i] = v64[i] = (uint32_t) i; k64[i] = k[i]; }
        glu_oracle_stable_sort_pairs_u32(k, v, n, 32);
        glu_oracle_stable_sort_pairs_u64(k64, v64, n, 64);
        int same = 1;
        for (uint64_t i = 0; i < n; i++) same = same && k64[i] == k[i] && v64[i] == v[i];
        EXPECT(same);
        free(k); free(v); free(k64); free(v64);
    }

    /* Blelloch scan (literal sweeps) against the plain exclusive scan, partitions of power-of-two length */
    for (uint64_t count = 1; count <= 4096; count *= 4)
        for (uint64_t parts = 1; parts <= 5; parts += 2)
        {
            uint32_t* d = malloc(count * parts * 4); uint32_t* e = malloc(count * parts * 4);
            glu_oracle_minstd_sample(123, count * parts, 0, 100, d);
            glu_oracle_exclusive_scan_u32(d, e, count, parts);
            EXPECT(glu_oracle_blelloch_scan_u32(d, count, parts) == 0);
            EXPECT(memcmp(d, e, count * parts * 4) == 0);
            free(d); free(e);
        }

    /* reduce (stride-32 recursion) against a loop, the reference's non-fitting sizes included */
    const uint64_t rsizes[] = {1, 31, 32, 93, 201, 693, 1024, 2087, 7358, 88289};
    for (unsigned i = 0; i < sizeof(rsizes) / sizeof(rsizes[0]); i++)
        for (int op = 0; op < 4; op++)
        {
            const uint64_t n = rsizes[i];
            uint32_t* d = malloc(n * 4);
            glu_oracle_minstd_sample(1, n, op == 1 ? 1 : 0, op == 1 ? 3 : 100, d);
            uint32_t expect = d[0];
            for (uint64_t j = 1; j < n; j++)
                expect = op == 0 ? expect + d[j] : op == 1 ? expect * d[j] : op == 2 ? (d[j] < expect ? d[j] : expect) : (d[j] > expect ? d[j] : expect);
            EXPECT(glu_oracle_reduce_reference_u32(d, n, op, 32) == 0); /* 32: the subgroup size the reference assumes (Reduce.hpp:26) */
            EXPECT(d[0] == expect);
            free(d);
        }
    printf("oracle selfcheck: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
