/*
 * error.c -- the reference's error convention [ref src/turtle/error.c:28-198,
 * error.h:31-110]: one process-global handler, called once per failing API
 * call with "{ <function> [#<code>], <file>:<line> } <text>"; the default
 * handler prints and exits; a NULL handler means silent return codes.
 */
#include "host.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void default_handler(
    enum turtle_return code, turtle_function_t * function, const char * message)
{
        (void)code;
        (void)function;
        fprintf(stderr, "A TURTLE library error occurred:\n%s\n", message);
        exit(EXIT_FAILURE);
}

static turtle_error_handler_t * g_handler = &default_handler;

turtle_error_handler_t * turtle_error_handler_get(void) { return g_handler; }

void turtle_error_handler_set(turtle_error_handler_t * handler)
{
        g_handler = handler;
}

enum turtle_return tamd_raise_(struct tamd_error * error, enum turtle_return rc,
    const char * file, int line, const char * format, ...)
{
        error->code = rc;
        if ((g_handler == NULL) || (rc == TURTLE_RETURN_SUCCESS)) return rc;

        char text[1024];
        const char * name = turtle_error_function(error->function);
        int used = snprintf(text, sizeof(text), "{ %s [#%d], %s:%d } ",
            name ? name : "(unknown)", (int)rc, file, line);
        if (used < 0) used = 0;
        if ((size_t)used < sizeof(text)) {
                va_list ap;
                va_start(ap, format);
                vsnprintf(text + used, sizeof(text) - used, format, ap);
                va_end(ap);
        }
        g_handler(rc, error->function, text);
        return rc;
}

const char * turtle_error_function(turtle_function_t * caller)
{
#define NAME(f)                                                                \
        if (caller == (turtle_function_t *)f) return #f
        NAME(turtle_client_clear);
        NAME(turtle_client_create);
        NAME(turtle_client_destroy);
        NAME(turtle_client_elevation);
        NAME(turtle_ecef_from_geodetic);
        NAME(turtle_ecef_from_horizontal);
        NAME(turtle_ecef_to_geodetic);
        NAME(turtle_ecef_to_horizontal);
        NAME(turtle_error_function);
        NAME(turtle_error_handler_get);
        NAME(turtle_error_handler_set);
        NAME(turtle_map_create);
        NAME(turtle_map_destroy);
        NAME(turtle_map_elevation);
        NAME(turtle_map_fill);
        NAME(turtle_map_gradient);
        NAME(turtle_map_gradient_n);
        NAME(turtle_stack_gradient);
        NAME(turtle_stack_gradient_n);
        NAME(turtle_map_dump);
        NAME(turtle_map_load);
        NAME(turtle_map_meta);
        NAME(turtle_map_node);
        NAME(turtle_map_projection);
        NAME(turtle_projection_configure);
        NAME(turtle_projection_create);
        NAME(turtle_projection_destroy);
        NAME(turtle_projection_name);
        NAME(turtle_projection_project);
        NAME(turtle_projection_unproject);
        NAME(turtle_projection_project_n);
        NAME(turtle_projection_unproject_n);
        NAME(turtle_stack_clear);
        NAME(turtle_stack_create);
        NAME(turtle_stack_destroy);
        NAME(turtle_stack_elevation);
        NAME(turtle_stack_load);
        NAME(turtle_stepper_add_flat);
        NAME(turtle_stepper_add_layer);
        NAME(turtle_stepper_add_map);
        NAME(turtle_stepper_add_stack);
        NAME(turtle_stepper_create);
        NAME(turtle_stepper_destroy);
        NAME(turtle_stepper_geoid_get);
        NAME(turtle_stepper_geoid_set);
        NAME(turtle_stepper_range_get);
        NAME(turtle_stepper_range_set);
        NAME(turtle_stepper_position);
        NAME(turtle_stepper_step);
        /* batch extension */
        NAME(turtle_ecef_from_geodetic_n);
        NAME(turtle_ecef_to_geodetic_n);
        NAME(turtle_ecef_from_horizontal_n);
        NAME(turtle_ecef_to_horizontal_n);
        NAME(turtle_map_elevation_n);
        NAME(turtle_stack_elevation_n);
        NAME(turtle_stepper_position_n);
        NAME(turtle_stepper_step_n);
        NAME(turtle_stepper_walk_n);
        NAME(turtle_stepper_trace_n);
        NAME(turtle_stepper_scatter_n);
        NAME(turtle_stepper_trace_stats);
        NAME(turtle_amd_tally_n);
        NAME(turtle_amd_philox_n);
        NAME(turtle_amd_isotropic_n);
        NAME(turtle_amd_device_set);
        NAME(turtle_amd_stream_set);
        NAME(turtle_amd_stepper_clone);
        NAME(turtle_amd_synchronize);
        return NULL;
#undef NAME
}
