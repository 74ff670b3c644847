/*
 * projection.c -- map projections: the handle, the name parser and the host
 * entry points [ref src/turtle/projection.c:53-230, projection.h:29-46].  The
 * formulas (Lambert conformal conic I-IV/IIe/93, UTM by the Krueger series)
 * are evaluated on the device (d_project / d_unproject in device.hip).
 */
#include "host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* [ref projection.c:83-95] */
static int locate_word(const char ** str)
{
        const char * p = *str;
        while (*p == ' ') p++;
        *str = p;
        int n = 0;
        while ((*p != ' ') && (*p != '\0')) p++, n++;
        return n;
}

/* [ref projection.c:98-171].  Returns 0 or an enum turtle_return with `message`
 * filled in.  The reference compares with strncmp over the length of the
 * word it found, so a prefix ("Lam", "U") is accepted and "Lambert" alone means
 * Lambert I: kept. */
int tamd_projection_configure(
    struct turtle_projection * projection, const char * name, char * message, size_t size)
{
        projection->type = TAMD_PROJ_NONE;
        projection->lambert_tag = 0;
        projection->longitude_0 = 0.;
        projection->hemisphere = 0;
        projection->tag[0] = 0x0;
        if (name == NULL) return TURTLE_RETURN_SUCCESS;

        const char * p = name;
        int n = locate_word(&p);
        if (n == 0) {
                snprintf(message, size, "missing projection specifier");
                return TURTLE_RETURN_BAD_PROJECTION;
        } else if (strncmp(p, "Lambert", n) == 0) {
                projection->type = TAMD_PROJ_LAMBERT;
                p += n;
                n = locate_word(&p);
                const char * tag[6] = { "I", "II", "IIe", "III", "IV", "93" };
                int i;
                for (i = 0; i < 6; i++) {
                        if (strncmp(p, tag[i], n) == 0) {
                                projection->lambert_tag = i;
                                goto done;
                        }
                }
        } else if (strncmp(p, "UTM", n) == 0) {
                projection->type = TAMD_PROJ_UTM;
                p += n;
                int zone;
                char hemisphere;
                if (sscanf(p, "%d%c", &zone, &hemisphere) != 2) {
                        snprintf(message, size, "invalid UTM specifier `%s'", p);
                        projection->type = TAMD_PROJ_NONE;
                        return TURTLE_RETURN_BAD_PROJECTION;
                }
                if (hemisphere == '.') {
                        double longitude_0;
                        if (sscanf(p, "%lf%c", &longitude_0, &hemisphere) != 2) {
                                snprintf(message, size, "invalid extended UTM specifier `%s'", p);
                                projection->type = TAMD_PROJ_NONE;
                                return TURTLE_RETURN_BAD_PROJECTION;
                        }
                        projection->longitude_0 = longitude_0;
                } else
                        projection->longitude_0 = 6. * zone - 183.;
                if (hemisphere == 'N')
                        projection->hemisphere = 1;
                else if (hemisphere == 'S')
                        projection->hemisphere = -1;
                else {
                        snprintf(message, size, "invalid UTM hemisphere `%c'", hemisphere);
                        projection->type = TAMD_PROJ_NONE;
                        return TURTLE_RETURN_BAD_PROJECTION;
                }
                goto done;
        }
        snprintf(message, size, "invalid projection `%s'", p);
        projection->type = TAMD_PROJ_NONE;
        return TURTLE_RETURN_BAD_PROJECTION;
done:
        strncpy(projection->tag, name, sizeof(projection->tag) - 1);
        projection->tag[sizeof(projection->tag) - 1] = 0x0;
        return TURTLE_RETURN_SUCCESS;
}

/* [ref projection.c:53-72] */
enum turtle_return turtle_projection_create(
    struct turtle_projection ** projection, const char * name)
{
        TAMD_ERROR_INIT(&turtle_projection_create);
        *projection = NULL;
        struct turtle_projection tmp;
        char message[256];
        const int rc = tamd_projection_configure(&tmp, name, message, sizeof(message));
        if (rc != TURTLE_RETURN_SUCCESS) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        *projection = malloc(sizeof(**projection));
        if (*projection == NULL)
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR, "could not allocate memory");
        memcpy(*projection, &tmp, sizeof(tmp));
        return TURTLE_RETURN_SUCCESS;
}

/* [ref projection.c:75-80] */
void turtle_projection_destroy(struct turtle_projection ** projection)
{
        if ((projection == NULL) || (*projection == NULL)) return;
        free(*projection);
        *projection = NULL;
}

/* [ref projection.c:173-179] */
enum turtle_return turtle_projection_configure(
    struct turtle_projection * projection, const char * name)
{
        TAMD_ERROR_INIT(&turtle_projection_configure);
        char message[256];
        const int rc = tamd_projection_configure(projection, name, message, sizeof(message));
        if (rc != TURTLE_RETURN_SUCCESS) return TAMD_RAISE((enum turtle_return)rc, "%s", message);
        return TURTLE_RETURN_SUCCESS;
}

/* [ref projection.c:182-189] */
const char * turtle_projection_name(const struct turtle_projection * projection)
{
        if ((projection == NULL) || (projection->type == TAMD_PROJ_NONE)) return NULL;
        return projection->tag;
}

void tamd_projection_desc(const struct turtle_projection * projection, struct tamd_proj * desc)
{
        desc->type = (projection != NULL) ? projection->type : TAMD_PROJ_NONE;
        desc->lambert_tag = (projection != NULL) ? projection->lambert_tag : 0;
        desc->longitude_0 = (projection != NULL) ? projection->longitude_0 : 0.;
        desc->hemisphere = (projection != NULL) ? projection->hemisphere : 0;
}

static int project_n(const struct turtle_projection * projection, int inverse, long n,
    const double * a, const double * b, double * c, double * d, int space)
{
        struct tamd_stage st;
        struct tamd_proj desc;
        void *da, *db, *dc, *dd;
        const size_t nb = (size_t)n * sizeof(double);
        tamd_projection_desc(projection, &desc);
        return tamd_stage_begin(&st, space, 4 * nb) || tamd_stage_in(&st, a, nb, &da) ||
            tamd_stage_in(&st, b, nb, &db) || tamd_stage_out(&st, c, nb, &dc) ||
            tamd_stage_out(&st, d, nb, &dd) || tamd_k_project(desc, inverse, n, da, db, dc, dd) ||
            tamd_stage_fetch(&st, c, nb, dc) || tamd_stage_fetch(&st, d, nb, dd) ||
            tamd_stage_end(&st);
}

static enum turtle_return check(struct tamd_error * error, const struct turtle_projection * p)
{
        struct tamd_error error_ = *error;
        if (p == NULL) return TAMD_RAISE(TURTLE_RETURN_BAD_ADDRESS, "missing projection");
        if (p->type == TAMD_PROJ_NONE)
                return TAMD_RAISE(TURTLE_RETURN_BAD_PROJECTION, "invalid projection");
        return TURTLE_RETURN_SUCCESS;
}

/* [ref projection.c:192-210] */
enum turtle_return turtle_projection_project(const struct turtle_projection * projection,
    double latitude, double longitude, double * x, double * y)
{
        TAMD_ERROR_INIT(&turtle_projection_project);
        *x = 0., *y = 0.;
        const enum turtle_return rc = check(&error_, projection);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        if (project_n(projection, 0, 1, &latitude, &longitude, x, y, TURTLE_AMD_HOST))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

/* [ref projection.c:213-230] */
enum turtle_return turtle_projection_unproject(const struct turtle_projection * projection,
    double x, double y, double * latitude, double * longitude)
{
        TAMD_ERROR_INIT(&turtle_projection_unproject);
        *latitude = 0., *longitude = 0.;
        const enum turtle_return rc = check(&error_, projection);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        if (project_n(projection, 1, 1, &x, &y, latitude, longitude, TURTLE_AMD_HOST))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_projection_project_n(const struct turtle_projection * projection,
    long n, const double * latitude, const double * longitude, double * x, double * y,
    int space)
{
        TAMD_ERROR_INIT(&turtle_projection_project_n);
        const enum turtle_return rc = check(&error_, projection);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        if (project_n(projection, 0, n, latitude, longitude, x, y, space))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}

enum turtle_return turtle_projection_unproject_n(const struct turtle_projection * projection,
    long n, const double * x, const double * y, double * latitude, double * longitude,
    int space)
{
        TAMD_ERROR_INIT(&turtle_projection_unproject_n);
        const enum turtle_return rc = check(&error_, projection);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        if (project_n(projection, 1, n, x, y, latitude, longitude, space))
                return TAMD_RAISE_DEVICE();
        return TURTLE_RETURN_SUCCESS;
}
