/*
 * paging.c -- host side of the paged tile stacks.
 *
 * A stack keeps at most `stack_size` tiles in memory [ref stack.c:150,
 * :399-450]; the reference loads a tile the moment a query needs it and drops
 * the least recently used one when the stack is full.  A batch on the GPU
 * cannot stop at a query, so the same policy runs in ROUNDS: the kernels list
 * the items (rays, points) that met a tile that is not resident, with a bitmap
 * of the tiles they want (device.hip, "Paging"); the host brings those tiles
 * into HBM -- dropping the least recently wanted ones beyond the budget --
 * and runs the list again, until it is empty.  An item is only ever computed
 * against tiles that are resident, so the results are those of a stack with
 * every tile in memory.
 *
 * This file owns the round bookkeeping: the two lists (one being read while
 * the next is written), their counters and the bitmap, all in one grow-only
 * device block of the calling thread.
 */
#include <stdlib.h>
#include <string.h>

#include "host.h"


int tamd_pager_begin(struct tamd_pager * pager, long n, int table_entries)
{
        memset(pager, 0, sizeof(*pager));
        if (n < 1) n = 1;
        const size_t list = (((size_t)n * sizeof(int) + 255) / 256) * 256;
        /* demand counters (one per tile-table entry), then the bitmap of the first item */
        const size_t counters = ((((size_t)table_entries + 1) * TAMD_DEMAND_STRIDE * sizeof(unsigned) + 255) / 256) * 256;
        const size_t bitmap = ((((size_t)table_entries + 31) / 32 + 1) * sizeof(unsigned) + 255) / 256 * 256;
        const size_t bytes = 2 * list + 256 + counters + bitmap;
        void * block;
        if (tamd_dev_block(0, &block, bytes, NULL)) return -1; /* the calling thread's */
        char * base = block;
        pager->d_list[0] = (int *)base;
        pager->d_list[1] = (int *)(base + list);
        pager->d_count = (unsigned long long *)(base + 2 * list); /* [0], [16]: a line apart */
        pager->d_wanted = (unsigned *)(base + 2 * list + 256);
        pager->words = (counters + bitmap) / sizeof(unsigned);
        pager->wanted = malloc(counters + bitmap);
        pager->pinned = calloc(1, bitmap);
        if ((pager->wanted == NULL) || (pager->pinned == NULL)) return -1;
        pager->wanted_first = pager->wanted + counters / sizeof(unsigned);
        pager->first_offset = counters / sizeof(unsigned);
        pager->bitmap_words = bitmap / sizeof(unsigned);
        pager->first_id = -1;
        pager->active = 1;
        return 0;
}

/* The paging block of the next round: reads the list the last round wrote (or
 * everything, the first time), writes the other one. */
int tamd_pager_round(struct tamd_pager * pager, struct tamd_paging * pg)
{
        memset(pg, 0, sizeof(*pg));
        pg->first_id = -1;
        if (!pager->active) return 0;
        pg->first_id = pager->first_id;
        const int out = pager->rounds & 1;
        if (pager->rounds > 0) {
                pg->ids = pager->d_list[out ^ 1];
                pg->n_in = pager->d_count + 16 * (out ^ 1);
        }
        pg->faulted = pager->d_list[out];
        pg->n_faulted = pager->d_count + 16 * out;
        pg->wanted = pager->d_wanted;
        pg->wanted_first = pager->d_wanted + pager->first_offset;
        if (tamd_dev_zero(pg->n_faulted, sizeof(*pg->n_faulted)) ||
            tamd_dev_zero(pager->d_wanted, pager->words * sizeof(unsigned)))
                return -1;
        return 0;
}

/* After the kernels of a round: how many items it listed, and (host copies) how
 * much each tile is wanted.  0 items: the batch is complete. */
int tamd_pager_collect(struct tamd_pager * pager, unsigned long long * n_faulted)
{
        *n_faulted = 0;
        if (!pager->active) return 0;
        const int out = pager->rounds & 1;
        if (tamd_dev_d2h(n_faulted, pager->d_count + 16 * out, sizeof(*n_faulted))) return -1;
        pager->rounds++;
        if (*n_faulted == 0) return 0;
        if (tamd_dev_d2h(pager->wanted, pager->d_wanted, pager->words * sizeof(unsigned))) return -1;
        /* The item served without fail.  It keeps every tile it has asked for
         * (`pinned`) until it stops waiting: a step that needs tiles A and B in turn
         * would otherwise see A go when B comes.  Then the first item of the
         * current list takes over. */
        size_t w;
        int waits = 0;
        for (w = 0; w < pager->bitmap_words; w++) waits |= (pager->wanted_first[w] != 0);
        if ((pager->first_id >= 0) && !waits) {
                pager->first_id = -1;
                memset(pager->pinned, 0, pager->bitmap_words * sizeof(unsigned));
        }
        if (pager->first_id < 0) {
                /* nobody was named: the kernel took the first of the list it wrote */
                if (tamd_dev_d2h(&pager->first_id, pager->d_list[out], sizeof(int))) return -1;
                if (!waits) pager->first_id = -1; /* cannot be: somebody is listed */
        }
        for (w = 0; w < pager->bitmap_words; w++) pager->pinned[w] |= pager->wanted_first[w];
        return 0;
}

void tamd_pager_end(struct tamd_pager * pager)
{
        free(pager->wanted), free(pager->pinned);
        pager->wanted = pager->pinned = NULL;
        pager->active = 0;
}
