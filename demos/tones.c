#include "tones.h"

#include <math.h>

void tones_init(tones_t *g) { g->count = 0; }

void tones_add(tones_t *g, double fsampl, double freq, double amp, double phase)
{
    struct tone *t = &g->t[g->count++];
    t->amp = amp;
    t->phase = phase;
    t->angle = 0.0;
    t->step = (2.0 * M_PI * freq) / fsampl; /* same expression as tonegen/tonegen.c:22 */
}

/* Each sample adds the tones in order and then advances each running angle by
 * its step: the recurrence of tonegen/tonegen.c:30-39, not a closed form. */
void tones_read(tones_t *g, double *buf, unsigned int n)
{
    for (unsigned int s = 0; s < n; s++) {
        double v = 0;
        for (unsigned int k = 0; k < g->count; k++) {
            struct tone *t = &g->t[k];
            v += t->amp * cos(t->angle + t->phase);
            t->angle += t->step;
        }
        buf[s] = v;
    }
}

void tones_read_complex(tones_t *g, double (*buf)[2], unsigned int n)
{
    for (unsigned int s = 0; s < n; s++) {
        double v = 0;
        for (unsigned int k = 0; k < g->count; k++) {
            struct tone *t = &g->t[k];
            v += t->amp * cos(t->angle + t->phase);
            t->angle += t->step;
        }
        buf[s][0] = v;
        buf[s][1] = 0;
    }
}

void tones_read_sin(tones_t *g, double *buf, unsigned int n)
{
    for (unsigned int s = 0; s < n; s++) {
        double v = 0;
        for (unsigned int k = 0; k < g->count; k++) {
            v += sin(g->t[k].angle);
            g->t[k].angle += g->t[k].step;
        }
        buf[s] = v;
    }
}
