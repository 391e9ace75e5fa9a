/* Host-side sum-of-tones generator for the demos: the role tonegen/tonegen.{c,h}
 * plays in the reference (same call order: init, add(fsampl, freq, a, phi), read).
 * Input synthesis runs on the host; it is not part of the device hot path. */
#ifndef DEMO_TONES_H_INCLUDED
#define DEMO_TONES_H_INCLUDED

#define TONES_MAX 32

struct tone { double amp, phase, angle, step; };
typedef struct tones { unsigned int count; struct tone t[TONES_MAX]; } tones_t;

void tones_init(tones_t *g);
void tones_add(tones_t *g, double fsampl, double freq, double amp, double phase);
void tones_read(tones_t *g, double *buf, unsigned int n);
void tones_read_complex(tones_t *g, double (*buf)[2], unsigned int n);
/* unit-amplitude sines from phase 0: the inline generator of fft/src/main.c:265-291 */
void tones_read_sin(tones_t *g, double *buf, unsigned int n);

#endif
