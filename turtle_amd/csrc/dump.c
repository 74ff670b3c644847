/*
 * dump.c -- turtle_map_dump [ref src/turtle/map.c:165-180]: a map written to
 * disk in one of the two formats the reference can write,
 *
 *   .png   its own map format [ref io/png16.c:456-545]: 16-bit greyscale, rows
 *          north->south, big-endian samples round((z - z0) / dz), and a tEXt
 *          chunk "Comment" holding {"topography" : {x0, y0, z0, x1, y1, z1 as
 *          C99 hex floats, "projection"}} -- written here without libpng
 *          (zlib's deflate and crc32 only; filter 0 on every row);
 *   .tif   GeoTIFF-16 [ref io/geotiff16.c:261-327]: uncompressed int16 samples
 *          round(z), ModelPixelScale and ModelTiepoint tags, for maps with the
 *          int16 z scale (z0 = -32767, dz = 1) and no projection -- written
 *          here without libtiff, as one little-endian strip.
 *
 * The other extensions the library reads (.hgt, .grd, .asc) cannot be written,
 * as in the reference [ref io/hgt.c:52-55, io/grd.c:53-56, io/asc.c:51-54].
 *
 * One deliberate difference: the reference's GeoTIFF writer puts the grid's
 * SOUTHERN row in scan line 0 [ref geotiff16.c:313-318] under a top-left tie
 * point, and its reader flips the rows [ref geotiff16.c:246-255], so that a map
 * it dumps comes back upside down (its own test uses a pattern that is
 * symmetric under the flip [ref tests/test-turtle.c:1093-1130]).  Here scan
 * line 0 is the northern row, as the tags say: a dump read back -- by this
 * library or by the reference -- is the map that was dumped.
 */
#include "host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static double node_value(const struct turtle_map * m, int ix, int iy)
{
        const uint16_t code = m->nodes[(size_t)iy * m->nx + ix];
        return m->is_signed ? (double)(int16_t)code : m->z0 + code * m->dz;
}

/* ---- png --------------------------------------------------------------- */

static void put32(unsigned char * b, uint32_t v)
{
        b[0] = (unsigned char)(v >> 24), b[1] = (unsigned char)(v >> 16);
        b[2] = (unsigned char)(v >> 8), b[3] = (unsigned char)v;
}

static int png_chunk(FILE * fid, const char * type, const unsigned char * data, size_t n)
{
        unsigned char head[8], tail[4];
        put32(head, (uint32_t)n);
        memcpy(head + 4, type, 4);
        uLong crc = crc32(0L, head + 4, 4);
        if (n > 0) crc = crc32(crc, data, (uInt)n);
        put32(tail, (uint32_t)crc);
        return (fwrite(head, 1, 8, fid) != 8) || ((n > 0) && (fwrite(data, 1, n, fid) != n)) ||
            (fwrite(tail, 1, 4, fid) != 4);
}

/* 0, or PATH_ERROR / MEMORY_ERROR / BAD_FORMAT (a write that failed) */
static int png_write(const char * path, const struct turtle_map * m)
{
        static const unsigned char signature[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
        const size_t nx = (size_t)m->nx, ny = (size_t)m->ny;
        const size_t row = 1 + 2 * nx, raw_size = row * ny;
        uLongf packed_size = compressBound((uLong)raw_size);
        unsigned char * raw = malloc(raw_size);
        unsigned char * packed = malloc(packed_size);
        int rc = TURTLE_RETURN_MEMORY_ERROR;
        FILE * fid = NULL;
        if ((raw == NULL) || (packed == NULL)) goto done;
        size_t i, j;
        for (i = 0; i < ny; i++) {
                unsigned char * p = raw + i * row;
                *p++ = 0; /* filter: none */
                for (j = 0; j < nx; j++) {
                        /* [ref png16.c:527-531] */
                        const double d =
                            round((node_value(m, (int)j, (int)(ny - 1 - i)) - m->z0) / m->dz);
                        const uint16_t v = (uint16_t)d;
                        *p++ = (unsigned char)(v >> 8), *p++ = (unsigned char)v;
                }
        }
        if (compress2(packed, &packed_size, raw, (uLong)raw_size, Z_DEFAULT_COMPRESSION) != Z_OK)
                goto done;

        rc = TURTLE_RETURN_PATH_ERROR;
        fid = fopen(path, "wb+");
        if (fid == NULL) goto done;
        rc = TURTLE_RETURN_BAD_FORMAT;
        unsigned char header[13];
        put32(header, (uint32_t)nx), put32(header + 4, (uint32_t)ny);
        header[8] = 16, header[9] = 0, header[10] = 0, header[11] = 0, header[12] = 0;
        /* [ref png16.c:489-508] */
        const char * name = turtle_projection_name(&m->projection);
        char text[2048];
        const int used = snprintf(text, sizeof(text),
            "Comment%c{\"topography\" : {\"x0\" : %a, \"y0\" : %a, \"z0\" : %a, \"x1\" : %a, "
            "\"y1\" : %a, \"z1\" : %a, \"projection\" : \"%s\"}}",
            0, m->x0, m->y0, m->z0, m->x0 + m->dx * (m->nx - 1), m->y0 + m->dy * (m->ny - 1),
            m->z0 + m->dz * 65535, (name == NULL) ? "" : name);
        if ((used < 0) || ((size_t)used >= sizeof(text))) goto done;
        if ((fwrite(signature, 1, 8, fid) != 8) || png_chunk(fid, "IHDR", header, 13) ||
            png_chunk(fid, "tEXt", (const unsigned char *)text, (size_t)used) ||
            png_chunk(fid, "IDAT", packed, packed_size) || png_chunk(fid, "IEND", NULL, 0))
                goto done;
        rc = TURTLE_RETURN_SUCCESS;
done:
        if ((fid != NULL) && (fclose(fid) != 0) && (rc == TURTLE_RETURN_SUCCESS))
                rc = TURTLE_RETURN_BAD_FORMAT;
        free(raw);
        free(packed);
        return rc;
}

/* ---- GeoTIFF-16 --------------------------------------------------------- */

static void le16(unsigned char * b, uint16_t v) { b[0] = (unsigned char)v, b[1] = (unsigned char)(v >> 8); }
static void le32(unsigned char * b, uint32_t v)
{
        le16(b, (uint16_t)v), le16(b + 2, (uint16_t)(v >> 16));
}
static void le_double(unsigned char * b, double v)
{
        uint64_t u;
        memcpy(&u, &v, 8);
        int i;
        for (i = 0; i < 8; i++) b[i] = (unsigned char)(u >> (8 * i));
}

static unsigned char * ifd_entry(unsigned char * e, int tag, int type, uint32_t count, uint32_t value)
{
        le16(e, (uint16_t)tag), le16(e + 2, (uint16_t)type), le32(e + 4, count);
        if ((type == 3) && (count == 1)) {
                le16(e + 8, (uint16_t)value), le16(e + 10, 0);
        } else
                le32(e + 8, value);
        return e + 12;
}

static int tiff_write(const char * path, const struct turtle_map * m)
{
        enum { N_TAGS = 14 };
        const size_t nx = (size_t)m->nx, ny = (size_t)m->ny;
        const uint32_t data_at = 8, data_size = (uint32_t)(2 * nx * ny);
        const uint32_t scale_at = data_at + data_size + (data_size & 1u);
        const uint32_t tie_at = scale_at + 24, ifd_at = tie_at + 48;
        unsigned char * rowbuf = malloc(2 * nx);
        if (rowbuf == NULL) return TURTLE_RETURN_MEMORY_ERROR;
        FILE * fid = fopen(path, "wb+");
        if (fid == NULL) {
                free(rowbuf);
                return TURTLE_RETURN_PATH_ERROR;
        }
        int rc = TURTLE_RETURN_BAD_FORMAT;
        unsigned char head[8] = { 'I', 'I', 42, 0 };
        le32(head + 4, ifd_at);
        if (fwrite(head, 1, 8, fid) != 8) goto done;
        size_t i, j;
        for (i = 0; i < ny; i++) { /* scan line i: the i-th row from the north */
                for (j = 0; j < nx; j++) {
                        /* [ref geotiff16.c:315-317] */
                        const double d = round(node_value(m, (int)j, (int)(ny - 1 - i)));
                        le16(rowbuf + 2 * j, (uint16_t)(int16_t)d);
                }
                if (fwrite(rowbuf, 2, nx, fid) != nx) goto done;
        }
        if ((data_size & 1u) && (fputc(0, fid) == EOF)) goto done;
        unsigned char doubles[72];
        le_double(doubles, m->dx), le_double(doubles + 8, m->dy), le_double(doubles + 16, 0.);
        le_double(doubles + 24, 0.), le_double(doubles + 32, 0.), le_double(doubles + 40, 0.);
        le_double(doubles + 48, m->x0); /* [ref geotiff16.c:296-299] */
        le_double(doubles + 56, m->y0 + (m->ny - 1) * m->dy), le_double(doubles + 64, 0.);
        if (fwrite(doubles, 1, 72, fid) != 72) goto done;
        unsigned char ifd[2 + 12 * N_TAGS + 4], * e = ifd + 2;
        le16(ifd, N_TAGS);
        e = ifd_entry(e, 256, 4, 1, (uint32_t)nx);       /* ImageWidth */
        e = ifd_entry(e, 257, 4, 1, (uint32_t)ny);       /* ImageLength */
        e = ifd_entry(e, 258, 3, 1, 16);                 /* BitsPerSample */
        e = ifd_entry(e, 259, 3, 1, 1);                  /* Compression: none */
        e = ifd_entry(e, 262, 3, 1, 1);                  /* Photometric: min is black */
        e = ifd_entry(e, 273, 4, 1, data_at);            /* StripOffsets */
        e = ifd_entry(e, 274, 3, 1, 1);                  /* Orientation: top left */
        e = ifd_entry(e, 277, 3, 1, 1);                  /* SamplesPerPixel */
        e = ifd_entry(e, 278, 4, 1, (uint32_t)ny);       /* RowsPerStrip */
        e = ifd_entry(e, 279, 4, 1, data_size);          /* StripByteCounts */
        e = ifd_entry(e, 284, 3, 1, 1);                  /* PlanarConfiguration: contiguous */
        e = ifd_entry(e, 296, 3, 1, 1);                  /* ResolutionUnit: none */
        e = ifd_entry(e, 33550, 12, 3, scale_at);        /* ModelPixelScale */
        e = ifd_entry(e, 33922, 12, 6, tie_at);          /* ModelTiepoint */
        le32(e, 0);
        if (fwrite(ifd, 1, sizeof(ifd), fid) != sizeof(ifd)) goto done;
        rc = TURTLE_RETURN_SUCCESS;
done:
        if ((fclose(fid) != 0) && (rc == TURTLE_RETURN_SUCCESS)) rc = TURTLE_RETURN_BAD_FORMAT;
        free(rowbuf);
        return rc;
}

/* ---- the API ------------------------------------------------------------ */

enum turtle_return turtle_map_dump(const struct turtle_map * map, const char * path)
{
        TAMD_ERROR_INIT(&turtle_map_dump);
        /* [ref io.c:80-103] */
        const char * ext = strrchr(path, '.');
        ext = (ext == NULL) ? "" : ext + 1;
        const int png = (strcmp(ext, "png") == 0), tif = (strcmp(ext, "tif") == 0);
        if (!png && !tif) {
                if ((strcmp(ext, "hgt") == 0) || (strcmp(ext, "grd") == 0) || (strcmp(ext, "asc") == 0))
                        return TAMD_RAISE(TURTLE_RETURN_BAD_FORMAT,
                            "invalid write format for file `%s'", path);
                return TAMD_RAISE(
                    TURTLE_RETURN_BAD_EXTENSION, "no valid format for file `%s'", path);
        }
        if (tif) { /* [ref geotiff16.c:266-278] */
                if ((map->z0 != -32767.) || (map->dz != 1.))
                        return TAMD_RAISE(TURTLE_RETURN_BAD_FORMAT,
                            "unsupported z scale when dumping map to `%s'", path);
                if (turtle_map_projection(map) != NULL)
                        return TAMD_RAISE(TURTLE_RETURN_BAD_FORMAT,
                            "unsupported projection when dumping map to `%s'", path);
        }
        const int rc = png ? png_write(path, map) : tiff_write(path, map);
        if (rc == TURTLE_RETURN_PATH_ERROR)
                return TAMD_RAISE(TURTLE_RETURN_PATH_ERROR, "could not open file `%s'", path);
        if (rc == TURTLE_RETURN_MEMORY_ERROR)
                return TAMD_RAISE(TURTLE_RETURN_MEMORY_ERROR,
                    "could not allocate memory when writing file `%s'", path);
        if (rc != TURTLE_RETURN_SUCCESS)
                return TAMD_RAISE(
                    TURTLE_RETURN_BAD_FORMAT, "an error occured when writing to file `%s'", path);
        return TURTLE_RETURN_SUCCESS;
}
