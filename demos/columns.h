/* Numeric text tables as the reference's command-line tools read them (util/common/csv.c:
 * one row per line, columns separated by one space, blank lines and lines starting with '#'
 * skipped, the column count taken from the first data line).  Values are kept column by
 * column so one column is a contiguous array. */
#ifndef COLUMNS_H_INCLUDED
#define COLUMNS_H_INCLUDED

#include <stddef.h>

typedef struct {
    size_t nrow, ncol;
    double *values; /* values[col * nrow + row] */
} columns_t;

/* 0 on success, -1 on any failure (unreadable file, no data line, short row). */
int columns_load(columns_t *t, const char *path);
void columns_free(columns_t *t);
/* column `col` of the table (col < ncol is the caller's business, as in the reference) */
static inline double *columns_get(const columns_t *t, size_t col) { return t->values + col * t->nrow; }

/* "0x.." -> hexadecimal integer, anything else -> strtod: the tools' number syntax
 * (util/fft/fft.c:169-174) */
double columns_number(const char *s);

#endif
