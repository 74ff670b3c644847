/*
 * client.c -- per-thread handle onto a locked stack [ref src/turtle/client.c:
 * 41-223].  The reference uses it to pin one tile per thread under the user's
 * lock; tiles here are immutable HBM residents shared by every launch, so a
 * client only forwards to its stack and returns the same answers.
 */
#include "host.h"

#include <stdlib.h>

enum turtle_return tamd_stack_elevation_scalar(struct turtle_stack * stack,
    turtle_function_t * caller, double latitude, double longitude, double * elevation,
    int * inside);

/* [ref client.c:41-66] */
enum turtle_return turtle_client_create(
    struct turtle_client ** client, struct turtle_stack * stack)
{
        TAMD_ERROR_INIT(&turtle_client_create);
        *client = NULL;
        if (stack == NULL)
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "invalid null stack");
        if (stack->lock == NULL)
                return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "stack has no lock");
        *client = malloc(sizeof(**client));
        if (*client == NULL)
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        (*client)->stack = stack;
        return TURTLE_RETURN_SUCCESS;
}

/* [ref client.c:69-90] */
enum turtle_return turtle_client_destroy(struct turtle_client ** client)
{
        if ((client == NULL) || (*client == NULL)) return TURTLE_RETURN_SUCCESS;
        free(*client);
        *client = NULL;
        return TURTLE_RETURN_SUCCESS;
}

/* [ref client.c:93-98]: nothing is pinned, so nothing to release */
enum turtle_return turtle_client_clear(struct turtle_client * client)
{
        (void)client;
        return TURTLE_RETURN_SUCCESS;
}

/* [ref client.c:101-188] */
enum turtle_return turtle_client_elevation(struct turtle_client * client,
    double latitude, double longitude, double * elevation, int * inside)
{
        return tamd_stack_elevation_scalar(client->stack,
            (turtle_function_t *)&turtle_client_elevation, latitude, longitude, elevation,
            inside);
}
