/*
 * text.c -- the two plain-text grid formats of the reference, used mostly for
 * geoid tables: .grd (EGM96 "ww15mgh.grd" layout) [ref src/turtle/io/grd.c:45-
 * 157] and ESRI .asc [ref src/turtle/io/asc.c:45-150].  Both quantise to 16
 * bits over the file's own [zmin, zmax] exactly as the reference does --
 * including its scan, which starts zmax at -DBL_MIN and only raises it in the
 * `else` of the zmin test [ref grd.c:88-106, asc.c:87-109].
 */
#include "host.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void text_meta_reset(struct turtle_map * m, const char * encoding)
{
        m->nx = m->ny = 0;
        m->x0 = m->y0 = m->z0 = 0., m->dx = m->dy = m->dz = 0.;
        m->is_signed = 0;
        m->projection.type = TAMD_PROJ_NONE;
        m->projection.tag[0] = 0x0;
        strcpy(m->encoding, encoding);
}

/* [ref grd.c:88-110, asc.c:87-113]; nodata = NAN disables the exclusion */
static int scan_range(FILE * fid, struct turtle_map * m, double nodata)
{
        double zmin = DBL_MAX, zmax = -DBL_MIN;
        long i;
        const long n = (long)m->nx * m->ny;
        for (i = 0; i < n; i++) {
                double d;
                if (fscanf(fid, "%lf", &d) != 1) return TURTLE_RETURN_BAD_FORMAT + 101;
                if (d == nodata)
                        continue;
                else if (d < zmin)
                        zmin = d;
                else if (d > zmax)
                        zmax = d;
        }
        m->z0 = zmin;
        m->dz = (zmax - zmin) / 65535;
        return TURTLE_RETURN_SUCCESS;
}

static uint16_t encode(const struct turtle_map * m, double z)
{
        return (uint16_t)round((z - m->z0) / m->dz); /* [ref grd.c:131-135] */
}

static int grd_header(FILE * fid, struct turtle_map * m)
{
        double h[6];
        if (fscanf(fid, "%lf %lf %lf %lf %lf %lf", h, h + 1, h + 2, h + 3, h + 4, h + 5) != 6)
                return TURTLE_RETURN_BAD_FORMAT + 102;
        m->x0 = h[2], m->dx = h[5], m->y0 = h[0], m->dy = h[4];
        m->nx = (int)round((h[3] - h[2]) / h[5]) + 1;
        m->ny = (int)round((h[1] - h[0]) / h[4]) + 1;
        return TURTLE_RETURN_SUCCESS;
}

int tamd_grd_probe(const char * path, struct turtle_map * m)
{
        text_meta_reset(m, "grd");
        FILE * fid = fopen(path, "r");
        if (fid == NULL) return TURTLE_RETURN_PATH_ERROR;
        int rc = grd_header(fid, m);
        if (rc == TURTLE_RETURN_SUCCESS) rc = scan_range(fid, m, NAN);
        fclose(fid);
        return rc;
}

/* [ref grd.c:137-157]: values in file order fill rows iy = 0, 1, ... */
int tamd_grd_read(const char * path, struct turtle_map * m)
{
        FILE * fid = fopen(path, "r");
        if (fid == NULL) return TURTLE_RETURN_PATH_ERROR;
        struct turtle_map header = *m;
        int rc = grd_header(fid, &header);
        if (rc == TURTLE_RETURN_SUCCESS) {
                const long n = (long)m->nx * m->ny;
                long i;
                for (i = 0; i < n; i++) {
                        double d;
                        if (fscanf(fid, "%lf", &d) != 1) break;
                        m->nodes[i] = encode(m, d);
                }
        }
        fclose(fid);
        return rc;
}

static int asc_header(FILE * fid, struct turtle_map * m, double * nodata)
{
        if ((fscanf(fid, "%*s %d", &m->nx) != 1) || (fscanf(fid, "%*s %d", &m->ny) != 1) ||
            (fscanf(fid, "%*s %lf", &m->x0) != 1) || (fscanf(fid, "%*s %lf", &m->y0) != 1) ||
            (fscanf(fid, "%*s %lf", &m->dx) != 1) || (fscanf(fid, "%*s %lf", nodata) != 1))
                return TURTLE_RETURN_BAD_FORMAT + 102;
        m->dy = m->dx; /* [ref asc.c:82-84]: cell corners -> cell centres */
        m->x0 += 0.5 * m->dx;
        m->y0 += 0.5 * m->dy;
        return TURTLE_RETURN_SUCCESS;
}

int tamd_asc_probe(const char * path, struct turtle_map * m)
{
        text_meta_reset(m, "asc");
        FILE * fid = fopen(path, "r");
        if (fid == NULL) return TURTLE_RETURN_PATH_ERROR;
        double nodata;
        int rc = asc_header(fid, m, &nodata);
        if (rc == TURTLE_RETURN_SUCCESS) rc = scan_range(fid, m, nodata);
        fclose(fid);
        return rc;
}

/* [ref asc.c:137-150]: file rows run north->south */
int tamd_asc_read(const char * path, struct turtle_map * m)
{
        FILE * fid = fopen(path, "r");
        if (fid == NULL) return TURTLE_RETURN_PATH_ERROR;
        struct turtle_map header = *m;
        double nodata;
        int rc = asc_header(fid, &header, &nodata);
        if (rc == TURTLE_RETURN_SUCCESS) {
                int ix, iy;
                for (iy = m->ny - 1; iy >= 0; iy--)
                        for (ix = 0; ix < m->nx; ix++) {
                                double d;
                                if (fscanf(fid, "%lf", &d) != 1) d = 0.;
                                m->nodes[(size_t)iy * m->nx + ix] = encode(m, d);
                        }
        }
        fclose(fid);
        return rc;
}
