/*
 * stage.c -- moves the arrays of a batch call between the caller's memory and
 * HBM.  TURTLE_AMD_DEVICE arrays are used in place (nothing is copied and the
 * call stays asynchronous); TURTLE_AMD_HOST arrays go through a grow-only
 * device arena of the calling thread and the call completes before it returns.
 *
 * Small HOST calls -- the scalar drop-in entry points above all: a dozen doubles
 * in, a dozen out -- go PACKED: the arrays are copied into a pinned buffer of the
 * thread and from there to the arena by asynchronous copies on the thread's
 * stream (ordered before the kernels, nothing to wait for), the outputs come back
 * the same way, and the ONE synchronisation of the call is in tamd_stage_end.
 * (Copy by copy, each with its own wait, a scalar turtle_stepper_step spent most
 * of its 20-40 us waiting.)
 */
#include "host.h"

#include <string.h>

#define TAMD_PACKED_BYTES ((size_t)256 * 1024)

static size_t round_up(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

int tamd_stage_begin(struct tamd_stage * st, int space, size_t total_bytes)
{
        st->space = space;
        st->packed = 0, st->n_pending = 0, st->pinned = NULL, st->pinned_used = 0;
        if (tamd_dev_init()) return 1;
        if (space == TURTLE_AMD_DEVICE) return 0;
        /* size the arena once, before any piece is handed out */
        void * all;
        tamd_scratch_reset();
        if (tamd_scratch_get(&all, total_bytes + 4096)) return 1;
        tamd_scratch_reset();
        if ((2 * total_bytes + 16 * 256 <= TAMD_PACKED_BYTES) &&
            (tamd_dev_pinned((void **)&st->pinned, TAMD_PACKED_BYTES) == 0))
                st->packed = 1;
        return 0;
}

static char * pinned_piece(struct tamd_stage * st, size_t bytes)
{
        if (st->pinned_used + round_up(bytes) > TAMD_PACKED_BYTES) return NULL;
        char * piece = st->pinned + st->pinned_used;
        st->pinned_used += round_up(bytes);
        return piece;
}

int tamd_stage_in(struct tamd_stage * st, const void * user, size_t bytes, void ** dev)
{
        if ((user == NULL) || (st->space == TURTLE_AMD_DEVICE)) {
                *dev = (void *)user;
                return 0;
        }
        if (tamd_scratch_get(dev, bytes)) return 1;
        char * piece = st->packed ? pinned_piece(st, bytes) : NULL;
        if (piece == NULL) return tamd_dev_h2d(*dev, user, bytes);
        memcpy(piece, user, bytes);
        return tamd_dev_copy_async(*dev, piece, bytes, 1);
}

int tamd_stage_out(struct tamd_stage * st, void * user, size_t bytes, void ** dev)
{
        if ((user == NULL) || (st->space == TURTLE_AMD_DEVICE)) {
                *dev = user;
                return 0;
        }
        return tamd_scratch_get(dev, bytes);
}

int tamd_stage_fetch(struct tamd_stage * st, void * user, size_t bytes, const void * dev)
{
        if ((user == NULL) || (st->space == TURTLE_AMD_DEVICE)) return 0;
        if (!st->packed || (st->n_pending >= TAMD_STAGE_PENDING)) return tamd_dev_d2h(user, dev, bytes);
        /* on its way back with the others: see tamd_stage_end */
        st->pending[st->n_pending].user = user, st->pending[st->n_pending].dev = dev;
        st->pending[st->n_pending].bytes = bytes;
        st->n_pending++;
        return 0;
}

int tamd_stage_end(struct tamd_stage * st)
{
        if (st->space == TURTLE_AMD_DEVICE) return 0;
        int i;
        if (st->n_pending > 0) {
                /* the outputs are pieces of one arena: ONE copy of the span they cover */
                const char *lo = st->pending[0].dev, *hi = lo;
                for (i = 0; i < st->n_pending; i++) {
                        const char * d = st->pending[i].dev;
                        if (d < lo) lo = d;
                        if (d + st->pending[i].bytes > hi) hi = d + st->pending[i].bytes;
                }
                char * piece = pinned_piece(st, (size_t)(hi - lo));
                if (piece == NULL) { /* (cannot be: the buffer holds twice the arena) */
                        for (i = 0; i < st->n_pending; i++)
                                if (tamd_dev_d2h(st->pending[i].user, st->pending[i].dev, st->pending[i].bytes))
                                        return 1;
                        st->n_pending = 0;
                        return tamd_dev_sync();
                }
                if (tamd_dev_copy_async(piece, lo, (size_t)(hi - lo), 0) || tamd_dev_sync()) return 1;
                for (i = 0; i < st->n_pending; i++)
                        memcpy(st->pending[i].user, piece + ((const char *)st->pending[i].dev - lo),
                            st->pending[i].bytes);
                st->n_pending = 0;
                return 0;
        }
        return tamd_dev_sync();
}
