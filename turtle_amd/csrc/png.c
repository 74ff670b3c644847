/*
 * png.c -- the reference's own map format, read without libpng [ref
 * src/turtle/io/png16.c:183-448, which goes through a dlopen()ed libpng]:
 * a 16-bit greyscale PNG whose samples are the 16-bit elevation codes
 * (big-endian, rows north->south, z = z0 + v dz [ref png16.c:405-410]) and a
 * tEXt chunk holding {"topography" : {"x0", "y0", "z0", "x1", "y1", "z1" (C99
 * hex floats), "projection" : "<name>"}} [ref png16.c:265-383, :497-505].
 *
 * Only zlib's inflate is borrowed; chunk walking, the five scan-line filters
 * and the header parser are here.  Interlaced images, palettes and other bit
 * depths are refused, as the reference refuses them [ref png16.c:240-253].
 */
#include "host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static uint32_t be32(const unsigned char * b)
{
        return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
}

/* value of `"key" :` inside text, as a number (strtod reads the %a form) */
static int json_number(const char * text, const char * key, double * value)
{
        char pattern[32];
        snprintf(pattern, sizeof(pattern), "\"%s\"", key);
        const char * p = strstr(text, pattern);
        if (p == NULL) return 1;
        p = strchr(p + strlen(pattern), ':');
        if (p == NULL) return 1;
        char * end;
        *value = strtod(p + 1, &end);
        return end == p + 1;
}

static int json_string(const char * text, const char * key, char * out, size_t size)
{
        char pattern[32];
        snprintf(pattern, sizeof(pattern), "\"%s\"", key);
        const char * p = strstr(text, pattern);
        if (p == NULL) return 1;
        p = strchr(p + strlen(pattern), ':');
        if (p == NULL) return 1;
        p = strchr(p, '"');
        if (p == NULL) return 1;
        const char * q = strchr(p + 1, '"');
        if ((q == NULL) || ((size_t)(q - p) > size)) return 1;
        memcpy(out, p + 1, q - p - 1);
        out[q - p - 1] = 0x0;
        return 0;
}

struct png_file {
        uint32_t width, height;
        unsigned char * idat; /* concatenated IDAT payloads */
        size_t idat_size;
        char * text;          /* the "topography" JSON, if any */
};

static void png_release(struct png_file * p)
{
        free(p->idat);
        free(p->text);
        p->idat = NULL, p->text = NULL;
}

/* Walk the chunks; with want_data == 0 stop collecting at the first IDAT */
static int png_open(const char * path, int want_data, struct png_file * p)
{
        static const unsigned char signature[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
        memset(p, 0, sizeof(*p));
        FILE * fid = fopen(path, "rb");
        if (fid == NULL) return TURTLE_RETURN_PATH_ERROR;
        unsigned char head[8];
        int rc = TURTLE_RETURN_BAD_FORMAT, seen_header = 0;
        if ((fread(head, 1, 8, fid) != 8) || (memcmp(head, signature, 8) != 0)) goto done;
        for (;;) {
                if (fread(head, 1, 8, fid) != 8) break;
                const uint32_t n = be32(head);
                const char * type = (const char *)head + 4;
                if (memcmp(type, "IHDR", 4) == 0) {
                        unsigned char h[13];
                        if ((n != 13) || (fread(h, 1, 13, fid) != 13)) goto done;
                        p->width = be32(h), p->height = be32(h + 4);
                        /* 16-bit greyscale, no interlace [ref png16.c:240-253] */
                        if ((h[8] != 16) || (h[9] != 0) || (h[12] != 0)) goto done;
                        seen_header = 1;
                        if (fseek(fid, 4, SEEK_CUR) != 0) goto done;
                } else if (memcmp(type, "tEXt", 4) == 0) {
                        char * body = malloc((size_t)n + 1);
                        if ((body == NULL) || (fread(body, 1, n, fid) != n)) {
                                free(body);
                                goto done;
                        }
                        body[n] = 0x0;
                        const size_t key = strlen(body) + 1; /* keyword, NUL, text */
                        if ((key <= n) && (strstr(body + key, "\"topography\"") != NULL) &&
                            (p->text == NULL)) {
                                memmove(body, body + key, n - key + 1);
                                p->text = body;
                        } else
                                free(body);
                        if (fseek(fid, 4, SEEK_CUR) != 0) goto done;
                } else if (memcmp(type, "IDAT", 4) == 0) {
                        if (!want_data) break;
                        unsigned char * grown = realloc(p->idat, p->idat_size + n);
                        if (grown == NULL) {
                                rc = TURTLE_RETURN_MEMORY_ERROR;
                                goto done;
                        }
                        p->idat = grown;
                        if (fread(p->idat + p->idat_size, 1, n, fid) != n) goto done;
                        p->idat_size += n;
                        if (fseek(fid, 4, SEEK_CUR) != 0) goto done;
                } else if (memcmp(type, "IEND", 4) == 0) {
                        break;
                } else if (fseek(fid, (long)n + 4, SEEK_CUR) != 0)
                        goto done;
        }
        if (seen_header && (p->width > 0) && (p->height > 0)) rc = TURTLE_RETURN_SUCCESS;
done:
        fclose(fid);
        if (rc != TURTLE_RETURN_SUCCESS) png_release(p);
        return rc;
}

int tamd_png_probe(const char * path, struct turtle_map * m)
{
        struct png_file p;
        int rc = png_open(path, 0, &p);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        m->nx = (int)p.width, m->ny = (int)p.height;
        m->x0 = m->y0 = m->z0 = 0., m->dx = m->dy = m->dz = 0.; /* [ref png16.c:204-206] */
        m->is_signed = 0;
        m->projection.type = TAMD_PROJ_NONE;
        m->projection.tag[0] = 0x0;
        strcpy(m->encoding, "png");
        if (p.text != NULL) { /* [ref png16.c:265-383] */
                double x1, y1, z1;
                char name[64], message[256];
                if (json_number(p.text, "x0", &m->x0) || json_number(p.text, "y0", &m->y0) ||
                    json_number(p.text, "z0", &m->z0) || json_number(p.text, "x1", &x1) ||
                    json_number(p.text, "y1", &y1) || json_number(p.text, "z1", &z1) ||
                    json_string(p.text, "projection", name, sizeof(name)))
                        rc = TURTLE_RETURN_BAD_FORMAT;
                else {
                        m->dx = (x1 - m->x0) / (m->nx - 1);
                        m->dy = (y1 - m->y0) / (m->ny - 1);
                        m->dz = (z1 - m->z0) / 65535;
                        if (name[0] != 0x0)
                                rc = tamd_projection_configure(
                                    &m->projection, name, message, sizeof(message));
                }
        }
        png_release(&p);
        return rc;
}

static unsigned char paeth(int a, int b, int c)
{
        const int p = a + b - c;
        const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
        return (unsigned char)(((pa <= pb) && (pa <= pc)) ? a : ((pb <= pc) ? b : c));
}

int tamd_png_read(const char * path, struct turtle_map * m)
{
        struct png_file p;
        int rc = png_open(path, 1, &p);
        if (rc != TURTLE_RETURN_SUCCESS) return rc;
        const size_t nx = p.width, ny = p.height, stride = 2 * nx, bpp = 2;
        const size_t raw_size = (stride + 1) * ny;
        unsigned char * raw = malloc(raw_size);
        rc = TURTLE_RETURN_BAD_FORMAT + 100; /* "missing data" unless all goes well */
        if (raw == NULL) {
                rc = TURTLE_RETURN_MEMORY_ERROR;
        } else {
                uLongf got = raw_size;
                if ((uncompress(raw, &got, p.idat, p.idat_size) == Z_OK) && (got == raw_size)) {
                        size_t r, i;
                        rc = TURTLE_RETURN_SUCCESS;
                        for (r = 0; r < ny; r++) {
                                unsigned char * line = raw + r * (stride + 1) + 1;
                                const unsigned char * up =
                                    (r > 0) ? raw + (r - 1) * (stride + 1) + 1 : NULL;
                                const int filter = line[-1];
                                for (i = 0; i < stride; i++) {
                                        const int a = (i >= bpp) ? line[i - bpp] : 0;
                                        const int b = up ? up[i] : 0;
                                        const int c = (up && (i >= bpp)) ? up[i - bpp] : 0;
                                        int add = 0;
                                        switch (filter) {
                                        case 0: break;
                                        case 1: add = a; break;
                                        case 2: add = b; break;
                                        case 3: add = (a + b) / 2; break;
                                        case 4: add = paeth(a, b, c); break;
                                        default: rc = TURTLE_RETURN_BAD_FORMAT; break;
                                        }
                                        line[i] = (unsigned char)(line[i] + add);
                                }
                                /* image row r (from the north) is grid row ny-1-r;
                                 * samples are big-endian [ref png16.c:405-410] */
                                uint16_t * dst = m->nodes + (ny - 1 - r) * nx;
                                for (i = 0; i < nx; i++)
                                        dst[i] = (uint16_t)((line[2 * i] << 8) | line[2 * i + 1]);
                        }
                }
        }
        free(raw);
        png_release(&p);
        return rc;
}
