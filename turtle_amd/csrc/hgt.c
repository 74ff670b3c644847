/*
 * hgt.c -- SRTM .hgt ingest [ref src/turtle/io/hgt.c:45-151].
 *
 * File layout: nx*ny big-endian int16, rows north->south.  The reference keeps
 * the payload raw and decodes in get_z on every node access; here the payload
 * is decoded ONCE into native-endian int16 with rows south->north, which is
 * the layout the kernels read.
 */
#include "host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Meta data from the file name only [ref io/hgt.c:59-104] */
int tamd_hgt_probe(const char * path, struct turtle_map * m)
{
        const char * filename = path;
        const char * p;
        for (p = path; *p != 0x0; p++)
                if ((*p == '/') || (*p == '\\')) filename = p + 1;
        if (strlen(filename) < 8) return TURTLE_RETURN_BAD_FORMAT;

        double x0 = atoi(filename + 4);
        if (filename[3] == 'W')
                x0 = -x0;
        else if (filename[3] != 'E')
                return TURTLE_RETURN_BAD_FORMAT;
        double y0 = atoi(filename + 1);
        if (filename[0] == 'S')
                y0 = -y0;
        else if (filename[0] != 'N')
                return TURTLE_RETURN_BAD_FORMAT;

        /* "N45E003.hgt" or "N45E003.SRTMGL1.hgt" => 3601 nodes, else 1201 */
        const char * ext = NULL;
        for (p = filename + 7; *p != 0x0; p++)
                if (*p == '.') ext = p + 1;
        if (ext == NULL) return TURTLE_RETURN_BAD_FORMAT;
        const int n = (int)(ext - filename) - 8;
        const int nodes =
            ((n == 0) || (strncmp(filename + 8, "SRTMGL1", n - 1) == 0)) ? 3601 : 1201;

        FILE * fid = fopen(path, "rb");
        if (fid == NULL) return TURTLE_RETURN_PATH_ERROR;
        fclose(fid);

        m->nx = m->ny = nodes;
        m->x0 = x0, m->y0 = y0;
        m->dx = 1. / (nodes - 1), m->dy = 1. / (nodes - 1);
        m->z0 = -32767., m->dz = 1.;
        m->is_signed = 1;
        m->projection.type = TAMD_PROJ_NONE;
        strcpy(m->encoding, "hgt"); /* [ref io.c:96-97] the extension */
        return TURTLE_RETURN_SUCCESS;
}

/* grid rows iy0 .. iy1 - 1 (file row r is grid row ny - 1 - r): the rows of a tile
 * are read band by band, a worker thread each (tiles.c) */
int tamd_hgt_read_rows(const char * path, struct turtle_map * m, int iy0, int iy1)
{
        FILE * fid = fopen(path, "rb");
        if (fid == NULL) return TURTLE_RETURN_PATH_ERROR;
        const size_t nx = m->nx, ny = m->ny;
        int rc = TURTLE_RETURN_SUCCESS;
        if (fseek(fid, (long)((ny - (size_t)iy1) * nx * sizeof(uint16_t)), SEEK_SET) != 0)
                rc = TURTLE_RETURN_BAD_FORMAT + 100;
        size_t r, i;
        for (r = ny - (size_t)iy1; (r < ny - (size_t)iy0) && (rc == TURTLE_RETURN_SUCCESS); r++) {
                uint16_t * dst = m->nodes + (ny - 1 - r) * nx;
                if (fread(dst, sizeof(*dst), nx, fid) != nx) {
                        rc = TURTLE_RETURN_BAD_FORMAT + 100; /* "missing data" */
                        break;
                }
                for (i = 0; i < nx; i++) { /* big-endian on file [ref io/hgt.c:127-131] */
                        const unsigned char * b = (const unsigned char *)&dst[i];
                        dst[i] = (uint16_t)((b[0] << 8) | b[1]);
                }
        }
        fclose(fid);
        return rc;
}

int tamd_hgt_read(const char * path, struct turtle_map * m)
{
        return tamd_hgt_read_rows(path, m, 0, m->ny);
}
