#include "columns.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    char *buf = NULL;
    long size;
    if (!f) return NULL;
    if (fseek(f, 0, SEEK_END) == 0 && (size = ftell(f)) >= 0 && fseek(f, 0, SEEK_SET) == 0) {
        buf = malloc((size_t)size + 1);
        if (buf && fread(buf, 1, (size_t)size, f) != (size_t)size) {
            free(buf);
            buf = NULL;
        }
        if (buf) {
            buf[size] = 0;
            *len = (size_t)size;
        }
    }
    fclose(f);
    return buf;
}

/* Next line holding data at or after *pos; returns its start and sets *end to its '\n' or the
 * end of the text.  NULL when none is left. */
static char *data_line(char *text, size_t len, size_t *pos, char **end)
{
    while (*pos < len) {
        char *start = text + *pos;
        char *nl = memchr(start, '\n', len - *pos);
        char *stop = nl ? nl : text + len;
        *pos = (size_t)(stop - text) + (nl ? 1 : 0);
        if (stop == start || *start == '#') continue;
        *end = stop;
        return start;
    }
    return NULL;
}

int columns_load(columns_t *t, const char *path)
{
    size_t len = 0, pos = 0, row;
    char *text = slurp(path, &len);
    char *line, *end;
    t->nrow = t->ncol = 0;
    t->values = NULL;
    if (!text) return -1;

    /* first pass: columns of the first data line (fields are runs up to a space), rows */
    if (!(line = data_line(text, len, &pos, &end))) goto fail;
    t->ncol = 1;
    for (; line != end; ++line)
        if (*line == ' ' && line + 1 != end) t->ncol++;
    for (t->nrow = 1; data_line(text, len, &pos, &end); t->nrow++) {}

    t->values = malloc(t->nrow * t->ncol * sizeof(double));
    if (!t->values) goto fail;

    /* second pass: ncol numbers from every data line */
    pos = 0;
    for (row = 0; (line = data_line(text, len, &pos, &end)); row++) {
        size_t col;
        char saved = *end;
        *end = 0; /* keep strtod inside the line */
        for (col = 0; col < t->ncol; col++) {
            char *after;
            if (line >= end) {
                *end = saved;
                goto fail;
            }
            columns_get(t, col)[row] = strtod(line, &after);
            line = after + 1; /* one separator after every value */
        }
        *end = saved;
    }
    free(text);
    return 0;
fail:
    free(text);
    columns_free(t);
    return -1;
}

void columns_free(columns_t *t)
{
    free(t->values);
    t->values = NULL;
    t->nrow = t->ncol = 0;
}

double columns_number(const char *s)
{
    if (strlen(s) > 2 && s[0] == '0' && s[1] == 'x') return (double)strtoul(s, NULL, 16);
    return strtod(s, NULL);
}
