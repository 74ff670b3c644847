/*
 * tiff.c -- GeoTIFF-16 ingest without libtiff [ref src/turtle/io/geotiff16.c:
 * 165-258, which reads through a dlopen()ed libtiff].
 *
 * Scope: what the reference itself writes and what SRTM/ASTER-GDEM elevation
 * tiles are: baseline TIFF, one 16-bit sample per pixel, uncompressed, in
 * strips, either byte order.  Compressed or tiled files are refused with
 * BAD_FORMAT.  Geo-referencing as the reference derives it: dx, dy from
 * ModelPixelScale (33550); x0 = tie point X, y0 = tie point Y + (1 - ny) dy
 * (33922) [ref geotiff16.c:205-214]; values are int16 elevations (z0 = -32767,
 * dz = 1, as [ref geotiff16.c:186-187, :230-233]); image rows run north->south
 * and are stored south->north in memory [ref geotiff16.c:246-255].
 */
#include "host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct tiff_file {
        FILE * fid;
        int swap; /* file byte order differs from the host's */
        uint32_t width, height, rows_per_strip, n_strips;
        uint32_t bits, samples, compression;
        uint32_t strip_offsets_at, strip_offsets_type, strip_offsets_count;
        uint32_t strip_offsets_value;
        double scale[3], tie[6];
        int have_scale, have_tie;
};

static uint16_t rd16(const unsigned char * b, int swap)
{
        uint16_t v;
        memcpy(&v, b, 2);
        return swap ? (uint16_t)((v >> 8) | (v << 8)) : v;
}

static uint32_t rd32(const unsigned char * b, int swap)
{
        uint32_t v;
        memcpy(&v, b, 4);
        if (swap) v = (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
        return v;
}

static double rd64f(const unsigned char * b, int swap)
{
        unsigned char t[8];
        int i;
        for (i = 0; i < 8; i++) t[i] = swap ? b[7 - i] : b[i];
        double v;
        memcpy(&v, t, 8);
        return v;
}

static int read_doubles(struct tiff_file * t, uint32_t offset, uint32_t count, double * out,
    uint32_t max)
{
        unsigned char buf[8];
        uint32_t i;
        if (count > max) count = max;
        if (fseek(t->fid, offset, SEEK_SET) != 0) return 1;
        for (i = 0; i < count; i++) {
                if (fread(buf, 1, 8, t->fid) != 8) return 1;
                out[i] = rd64f(buf, t->swap);
        }
        return 0;
}

/* Parse the header and the first image file directory */
static int tiff_open(const char * path, struct tiff_file * t)
{
        memset(t, 0, sizeof(*t));
        t->bits = 1, t->samples = 1, t->compression = 1;
        t->rows_per_strip = 0xffffffffu;
        t->fid = fopen(path, "rb");
        if (t->fid == NULL) return TURTLE_RETURN_PATH_ERROR;
        unsigned char h[8];
        if (fread(h, 1, 8, t->fid) != 8) goto bad;
        const uint16_t probe = 1;
        const int host_little = *(const unsigned char *)&probe;
        if ((h[0] == 'I') && (h[1] == 'I'))
                t->swap = !host_little;
        else if ((h[0] == 'M') && (h[1] == 'M'))
                t->swap = host_little;
        else
                goto bad;
        if (rd16(h + 2, t->swap) != 42) goto bad;
        const uint32_t ifd = rd32(h + 4, t->swap);
        if (fseek(t->fid, ifd, SEEK_SET) != 0) goto bad;
        unsigned char nb[2];
        if (fread(nb, 1, 2, t->fid) != 2) goto bad;
        const uint16_t n = rd16(nb, t->swap);
        uint16_t i;
        for (i = 0; i < n; i++) {
                unsigned char e[12];
                if (fseek(t->fid, ifd + 2 + 12u * i, SEEK_SET) != 0) goto bad;
                if (fread(e, 1, 12, t->fid) != 12) goto bad;
                const uint16_t tag = rd16(e, t->swap), type = rd16(e + 2, t->swap);
                const uint32_t count = rd32(e + 4, t->swap);
                /* SHORT values sit left-justified in the value field */
                const uint32_t value =
                    (type == 3) ? rd16(e + 8, t->swap) : rd32(e + 8, t->swap);
                switch (tag) {
                case 256: t->width = value; break;
                case 257: t->height = value; break;
                case 258: t->bits = value; break;
                case 259: t->compression = value; break;
                case 277: t->samples = value; break;
                case 278: t->rows_per_strip = value; break;
                case 273:
                        t->strip_offsets_type = type;
                        t->strip_offsets_count = count;
                        t->strip_offsets_value = value;
                        t->strip_offsets_at = rd32(e + 8, t->swap);
                        break;
                case 322: /* TileWidth: a tiled file */
                        goto bad;
                case 33550:
                        if ((type == 12) && (count >= 2)) {
                                if (read_doubles(t, rd32(e + 8, t->swap), count, t->scale, 3))
                                        goto bad;
                                t->have_scale = 1;
                        }
                        break;
                case 33922:
                        if ((type == 12) && (count >= 6)) {
                                if (read_doubles(t, rd32(e + 8, t->swap), count, t->tie, 6))
                                        goto bad;
                                t->have_tie = 1;
                        }
                        break;
                default: break;
                }
        }
        if ((t->width == 0) || (t->height == 0) || (t->bits != 16) || (t->samples != 1) ||
            (t->compression != 1) || (t->strip_offsets_count == 0))
                goto bad;
        if (t->rows_per_strip > t->height) t->rows_per_strip = t->height;
        t->n_strips = (t->height + t->rows_per_strip - 1) / t->rows_per_strip;
        if (t->n_strips != t->strip_offsets_count) goto bad;
        return TURTLE_RETURN_SUCCESS;
bad:
        fclose(t->fid);
        t->fid = NULL;
        return TURTLE_RETURN_BAD_FORMAT;
}

static int strip_offset(struct tiff_file * t, uint32_t strip, uint32_t * offset)
{
        if (t->strip_offsets_count == 1) {
                *offset = t->strip_offsets_value;
                return 0;
        }
        const uint32_t size = (t->strip_offsets_type == 3) ? 2 : 4;
        if ((t->strip_offsets_count == 2) && (size == 2)) { /* two SHORTs inline */
                *offset = (strip == 0) ? (t->strip_offsets_at & 0xffffu) : (t->strip_offsets_at >> 16);
                return 0;
        }
        unsigned char b[4];
        if (fseek(t->fid, t->strip_offsets_at + size * strip, SEEK_SET) != 0) return 1;
        if (fread(b, 1, size, t->fid) != size) return 1;
        *offset = (size == 2) ? rd16(b, t->swap) : rd32(b, t->swap);
        return 0;
}

int tamd_tiff_probe(const char * path, struct turtle_map * m)
{
        struct tiff_file t;
        const int rc = tiff_open(path, &t);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        fclose(t.fid);
        m->nx = (int)t.width, m->ny = (int)t.height;
        m->x0 = m->y0 = 0., m->dx = m->dy = 0.;
        if (t.have_scale) m->dx = t.scale[0], m->dy = t.scale[1];
        if (t.have_tie) {
                m->x0 = t.tie[3];
                m->y0 = t.tie[4] + (1 - m->ny) * m->dy; /* [ref geotiff16.c:213] */
        }
        m->z0 = -32767., m->dz = 1.;
        m->is_signed = 1;
        m->projection.type = TAMD_PROJ_NONE;
        strcpy(m->encoding, "tif");
        return TURTLE_RETURN_SUCCESS;
}

/* grid rows iy0 .. iy1 - 1 (image row `row`, from the north, is grid row ny - 1 - row) */
int tamd_tiff_read_rows(const char * path, struct turtle_map * m, int iy0, int iy1)
{
        struct tiff_file t;
        int rc = tiff_open(path, &t);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        const size_t nx = t.width;
        uint32_t row, in_strip = 0xffffffffu;
        for (row = t.height - (uint32_t)iy1; (row < t.height - (uint32_t)iy0) && (rc == TURTLE_RETURN_SUCCESS);
             row++) {
                const uint32_t strip = row / t.rows_per_strip;
                if (strip != in_strip) {
                        uint32_t offset;
                        if (strip_offset(&t, strip, &offset) ||
                            (fseek(t.fid, (long)(offset + (size_t)(row % t.rows_per_strip) * nx * sizeof(uint16_t)),
                                 SEEK_SET) != 0)) {
                                rc = TURTLE_RETURN_BAD_FORMAT + 100;
                                break;
                        }
                        in_strip = strip;
                }
                uint16_t * dst = m->nodes + ((size_t)t.height - 1 - row) * nx;
                if (fread(dst, sizeof(*dst), nx, t.fid) != nx) {
                        rc = TURTLE_RETURN_BAD_FORMAT + 100;
                        break;
                }
                if (t.swap) {
                        size_t i;
                        for (i = 0; i < nx; i++)
                                dst[i] = (uint16_t)((dst[i] >> 8) | (dst[i] << 8));
                }
        }
        fclose(t.fid);
        return rc;
}

int tamd_tiff_read(const char * path, struct turtle_map * m)
{
        return tamd_tiff_read_rows(path, m, 0, m->ny);
}
