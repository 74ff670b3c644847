/*
 * stage.c -- moves the arrays of a batch call between the caller's memory and
 * HBM.  TURTLE_AMD_DEVICE arrays are used in place (nothing is copied and the
 * call stays asynchronous); TURTLE_AMD_HOST arrays go through a grow-only
 * device arena and the call completes before it returns.
 */
#include "host.h"

int tamd_stage_begin(struct tamd_stage * st, int space, size_t total_bytes)
{
        st->space = space;
        if (tamd_dev_init()) return 1;
        if (space == TURTLE_AMD_DEVICE) return 0;
        /* size the arena once, before any piece is handed out */
        void * all;
        tamd_scratch_reset();
        if (tamd_scratch_get(&all, total_bytes + 4096)) return 1;
        tamd_scratch_reset();
        return 0;
}

int tamd_stage_in(struct tamd_stage * st, const void * user, size_t bytes, void ** dev)
{
        if ((user == NULL) || (st->space == TURTLE_AMD_DEVICE)) {
                *dev = (void *)user;
                return 0;
        }
        if (tamd_scratch_get(dev, bytes)) return 1;
        return tamd_dev_h2d(*dev, user, bytes);
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
        return tamd_dev_d2h(user, dev, bytes);
}

int tamd_stage_end(struct tamd_stage * st)
{
        if (st->space == TURTLE_AMD_DEVICE) return 0;
        return tamd_dev_sync();
}
