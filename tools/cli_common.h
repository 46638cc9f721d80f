/* cli_common.h -- what the command-line drivers (tools/herest.c, tools/hvite.c) share: HTK-style option scanning, the configuration
 * file, parameter kinds, file-name composition, and loading a batch of parameter files into one device table of observations.
 * Host code over include/htk_amd.h only (the drivers are the programs a user of the reference's HERest / HVite switches to;
 * flags follow HTKBook ref.tex "HERest" / "HVite" for the subset SURVEY.md 8(b) lists).
 */
#ifndef HTKAMD_CLI_COMMON_H
#define HTKAMD_CLI_COMMON_H

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "htk_amd.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "ERROR [%d] %s: %s\n", rc_, #call, htkamd_last_error()); exit(1); } } while (0)
#define DIE(...) do { fprintf(stderr, "ERROR "); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

/* ---- string list ---- */
typedef struct { char **v; int n, cap; } strlist;
static void sl_add(strlist *l, const char *s)
{
   if (l->n + 1 > l->cap) { l->cap = l->cap * 2 + 64; l->v = (char **)realloc(l->v, sizeof(char *) * (size_t)l->cap); }
   l->v[l->n++] = strdup(s);
}

/* ---- configuration file: NAME = value lines, optional MODULE: prefix, # comments (ReadConfigFile HShell.c:392) ---- */
typedef struct { strlist key, val, mod; } config;
static void cfg_read(config *c, const char *path)
{
   FILE *f = fopen(path, "r");
   char line[2048];
   if (!f) DIE("cannot open configuration file %s", path);
   while (fgets(line, sizeof(line), f)) {
      char *p = line, *eq, *k, *v, *e;
      char *hash = strchr(p, '#'); if (hash) *hash = 0;
      eq = strchr(p, '=');
      if (!eq) continue;
      *eq = 0;
      k = p; while (isspace((unsigned char)*k)) k++;
      e = k + strlen(k); while (e > k && isspace((unsigned char)e[-1])) *--e = 0;
      char *modname = (char *)"";
      { char *colon = strrchr(k, ':'); if (colon) { *colon = 0; modname = k; k = colon + 1; while (isspace((unsigned char)*k)) k++; } }   /* HPARM: TARGETKIND -> TARGETKIND */
      for (char *q = modname; *q; q++) *q = (char)toupper((unsigned char)*q);
      { char *e2 = modname + strlen(modname); while (e2 > modname && isspace((unsigned char)e2[-1])) *--e2 = 0; }
      v = eq + 1; while (isspace((unsigned char)*v)) v++;
      e = v + strlen(v); while (e > v && isspace((unsigned char)e[-1])) *--e = 0;
      for (char *q = k; *q; q++) *q = (char)toupper((unsigned char)*q);
      sl_add(&c->key, k); sl_add(&c->val, v); sl_add(&c->mod, modname);
   }
   fclose(f);
}
static const char *cfg_get(const config *c, const char *key)
{
   for (int i = c->key.n - 1; i >= 0; i--) if (!strcmp(c->key.v[i], key)) return c->val.v[i];
   return NULL;
}
/* a module's own parameter: `MODULE: NAME = v` or an unqualified `NAME = v` (GetConfig(name, incGlob = TRUE) HShell.c:601) */
__attribute__((unused)) static const char *cfg_get_mod(const config *c, const char *module, const char *key)
{
   for (int i = c->key.n - 1; i >= 0; i--)
      if (!strcmp(c->key.v[i], key) && (c->mod.v[i][0] == 0 || !strcmp(c->mod.v[i], module))) return c->val.v[i];
   return NULL;
}
static int cfg_int(const config *c, const char *key, int dflt) { const char *v = cfg_get(c, key); return v ? atoi(v) : dflt; }
static int cfg_bool(const config *c, const char *key, int dflt) { const char *v = cfg_get(c, key); return v ? (v[0] == 'T' || v[0] == 't') : dflt; }

/* ---- parameter kinds (HParm.h:40-75): base code in the low 6 bits, qualifier bits above ---- */
#define PK_HASENERGY 0100
#define PK_HASNULLE  0200
#define PK_HASDELTA  0400
#define PK_HASACCS   01000
#define PK_HASZEROM  04000
#define PK_HASZEROC  020000
#define PK_HASTHIRD  0100000
static const char *const pk_base[] = {"WAVEFORM", "LPC", "LPREFC", "LPCEPSTRA", "LPDELCEP", "IREFC", "MFCC", "FBANK", "MELSPEC", "USER", "DISCRETE", "PLP", NULL};
static int kind_parse(const char *s)
{
   char buf[128]; int k = -1;
   snprintf(buf, sizeof(buf), "%s", s);
   for (char *q = buf; *q; q++) *q = (char)toupper((unsigned char)*q);
   char *tok = strtok(buf, "_");
   for (int i = 0; pk_base[i]; i++) if (tok && !strcmp(tok, pk_base[i])) k = i;
   if (k < 0) DIE("unknown parameter kind %s", s);
   while ((tok = strtok(NULL, "_")) != NULL) {
      switch (tok[0]) {
      case 'E': k |= PK_HASENERGY; break; case 'N': k |= PK_HASNULLE; break; case 'D': k |= PK_HASDELTA; break; case 'A': k |= PK_HASACCS; break;
      case 'Z': k |= PK_HASZEROM; break; case '0': k |= PK_HASZEROC; break; case 'T': k |= PK_HASTHIRD; break;
      case 'C': case 'K': break;                               /* storage qualifiers: no effect on the observation */
      default: DIE("unknown qualifier _%s in parameter kind %s", tok, s);
      }
   }
   return k;
}

/* ---- file names: MakeFN (HShell.c:1258): directory and extension of `fn` replaced ---- */
static void make_fn(const char *fn, const char *dir, const char *ext, char *out, size_t n)
{
   const char *base = strrchr(fn, '/'); base = base ? base + 1 : fn;
   char stem[1024];
   snprintf(stem, sizeof(stem), "%s", base);
   if (ext) { char *dot = strrchr(stem, '.'); if (dot) *dot = 0; }
   if (dir) snprintf(out, n, "%s/%s%s%s", dir, stem, ext ? "." : "", ext ? ext : "");
   else if (ext) { char d2[1024]; snprintf(d2, sizeof(d2), "%.*s", (int)(base - fn), fn); snprintf(out, n, "%s%s.%s", d2, stem, ext); }
   else snprintf(out, n, "%s", fn);
}

/* ---- a batch of parameter files -> one device table of observations of the TARGET kind ----
 * The files hold `fileKind` (all the same); the qualifiers the target kind has beyond it (_D _A _T _Z _N) are computed on the device
 * (htkamd_parm_qualify = AddQualifiers HParm.c:1618), as OpenBuffer does when TARGETKIND asks for more than the file has. */
typedef struct {
   int nUtt, cols, period, *frameOff;
   float *dX;                  /* device [frameOff[nUtt] * cols] */
} obs_batch;

static double cfg_flt(const config *c, const char *key, double dflt) { const char *v = cfg_get(c, key); return v ? atof(v) : dflt; }

/* Waveform sources (SOURCEFORMAT = WAV, or SOURCEKIND = WAVEFORM with HTK waveform files): the files of the batch are coded on the
   device as OpenBuffer does for a waveform file (htkamd_mfcc_compute: statics + _0 / _E with HParm's configuration variables and
   their defaults, HParm.c:337-367); the differentials, _Z and _N of TARGETKIND follow in load_observations as for parameter files.
   Returns the parameter kind of the table (MFCC + _0 / _E). */
static int code_waveforms(const strlist *files, int first, int count, int targetKind, const config *cfg, obs_batch *ob, float **dStatOut, int *nStatOut)
{
   const char *sfmt = cfg_get(cfg, "SOURCEFORMAT");
   const int fmt = (sfmt && !strcasecmp(sfmt, "WAV")) ? HTKAMD_WAVE_WAV : HTKAMD_WAVE_HTK;
   if (targetKind < 0 || (targetKind & 077) != 6) DIE("waveform sources are coded as MFCC: TARGETKIND = MFCC[_0][_E][_D][_A][_T][_Z][_N] expected");
   short *all = NULL; size_t cap = 0;
   int *sampOff = (int *)calloc((size_t)count + 1, sizeof(int));
   double period = cfg_flt(cfg, "SOURCERATE", 0.0);
   for (int u = 0; u < count; u++) {
      short *x; long n; double per;
      CHECK(htkamd_wave_read(files->v[first + u], fmt, &x, &n, &per));
      if (period <= 0.0) period = per;                               /* SOURCERATE, when set, replaces the header's value (HParm.c:3940) */
      const size_t need = (size_t)sampOff[u] + (size_t)n;
      if (need > cap) { cap = need * 2 + 4096; all = (short *)realloc(all, sizeof(short) * cap); }
      memcpy(all + sampOff[u], x, sizeof(short) * (size_t)n);
      sampOff[u + 1] = sampOff[u] + (int)n;
      htkamd_free(x);
   }
   htkamd_mfcc_config mc; memset(&mc, 0, sizeof(mc));
   mc.sampPeriod = period; mc.winDur = cfg_flt(cfg, "WINDOWSIZE", 256000.0); mc.frPeriod = cfg_flt(cfg, "TARGETRATE", 100000.0);
   mc.numChans = cfg_int(cfg, "NUMCHANS", 20); mc.numCeps = cfg_int(cfg, "NUMCEPS", 12); mc.cepLifter = cfg_int(cfg, "CEPLIFTER", 22);
   mc.preEmph = (float)cfg_flt(cfg, "PREEMCOEF", 0.97); mc.useHam = cfg_bool(cfg, "USEHAMMING", 1); mc.usePower = cfg_bool(cfg, "USEPOWER", 0);
   mc.zMeanSource = cfg_bool(cfg, "ZMEANSOURCE", 0); mc.rawEnergy = cfg_bool(cfg, "RAWENERGY", 1); mc.eNormalise = cfg_bool(cfg, "ENORMALISE", 1);
   mc.loFreq = (float)cfg_flt(cfg, "LOFREQ", -1.0); mc.hiFreq = (float)cfg_flt(cfg, "HIFREQ", -1.0); mc.cepScale = (float)cfg_flt(cfg, "CEPSCALE", 1.0);
   mc.silFloor = (float)cfg_flt(cfg, "SILFLOOR", 50.0); mc.eScale = (float)cfg_flt(cfg, "ESCALE", 0.1);
   mc.hasC0 = (targetKind & PK_HASZEROC) != 0; mc.hasE = (targetKind & PK_HASENERGY) != 0;
   mc.delWin = 2; mc.accWin = 2;
   htkamd_mfcc *fe; CHECK(htkamd_mfcc_create(&mc, &fe));
   int F = 0;
   for (int u = 0; u < count; u++) F += htkamd_mfcc_num_frames(&mc, sampOff[u + 1] - sampOff[u]);
   const int cols = htkamd_mfcc_num_cols(&mc);
   short *dWav; float *dStat;
   CHECK(htkamd_dev_malloc((void **)&dWav, sizeof(short) * (size_t)(sampOff[count] ? sampOff[count] : 1)));
   CHECK(htkamd_memcpy_h2d(dWav, all, sizeof(short) * (size_t)sampOff[count], NULL));
   CHECK(htkamd_dev_malloc((void **)&dStat, sizeof(float) * (size_t)(F ? F : 1) * cols));
   CHECK(htkamd_mfcc_compute(fe, dWav, sampOff, count, ob->frameOff, dStat, NULL));
   CHECK(htkamd_stream_sync(NULL));
   CHECK(htkamd_dev_free(dWav)); htkamd_mfcc_destroy(fe); free(all); free(sampOff);
   ob->period = (int)(mc.frPeriod + 0.5);
   *dStatOut = dStat; *nStatOut = cols;
   return 6 | (mc.hasC0 ? PK_HASZEROC : 0) | (mc.hasE ? PK_HASENERGY : 0);
}

static void load_observations(const strlist *files, int first, int count, int targetKind, const config *cfg, obs_batch *ob)
{
   float *stat = NULL; size_t cap = 0;
   int nStat = 0, fileKind = -1;
   float *dStat = NULL;
   ob->nUtt = count; ob->frameOff = (int *)calloc((size_t)count + 1, sizeof(int)); ob->period = 100000;
   const char *sfmtL = cfg_get(cfg, "SOURCEFORMAT"), *skindL = cfg_get(cfg, "SOURCEKIND");
   const int waveform = (sfmtL && !strcasecmp(sfmtL, "WAV")) || (skindL && !strcasecmp(skindL, "WAVEFORM"));
   if (waveform) fileKind = code_waveforms(files, first, count, targetKind, cfg, ob, &dStat, &nStat);
   else
   for (int u = 0; u < count; u++) {
      float *x; int T, cols, pk, per;
      CHECK(htkamd_parm_read(files->v[first + u], &x, &T, &cols, &per, &pk));
      if (u == 0) { nStat = cols; fileKind = pk; ob->period = per; }
      if (cols != nStat || pk != fileKind) DIE("%s: kind/width differs from the first file of the batch", files->v[first + u]);
      const size_t need = (size_t)(ob->frameOff[u] + T) * nStat;
      if (need > cap) { cap = need * 2 + 4096; stat = (float *)realloc(stat, sizeof(float) * cap); }
      memcpy(stat + (size_t)ob->frameOff[u] * nStat, x, sizeof(float) * (size_t)T * nStat);
      ob->frameOff[u + 1] = ob->frameOff[u] + T;
      htkamd_free(x);
   }
   if (targetKind < 0) targetKind = fileKind;
   if ((targetKind & 077) != (fileKind & 077)) DIE("files hold base kind %s, TARGETKIND wants %s", pk_base[fileKind & 077], pk_base[targetKind & 077]);
   const int add = targetKind & ~fileKind, lost = fileKind & ~targetKind;
   if (lost & ~PK_HASNULLE) DIE("TARGETKIND drops qualifiers the files have (0%o)", lost);
   if (add & (PK_HASENERGY | PK_HASZEROC)) DIE("TARGETKIND asks for _E / _0, which cannot be derived from parameter files");
   const int F = ob->frameOff[count];
   if (!waveform) {
      CHECK(htkamd_dev_malloc((void **)&dStat, sizeof(float) * (size_t)(F ? F : 1) * nStat));
      CHECK(htkamd_memcpy_h2d(dStat, stat, sizeof(float) * (size_t)F * nStat, NULL));
      free(stat);
   }
   if (add == 0) { ob->dX = dStat; ob->cols = nStat; return; }
   if (fileKind & (PK_HASDELTA | PK_HASACCS | PK_HASTHIRD)) DIE("files already carry differentials: further qualifiers cannot be appended");
   htkamd_parm_quals q; memset(&q, 0, sizeof(q));
   const int nE = ((fileKind & PK_HASENERGY) ? 1 : 0) + ((fileKind & PK_HASZEROC) ? 1 : 0), base = nStat - nE;
   q.nStat = nStat;
   q.hasD = (targetKind & PK_HASDELTA) != 0; q.hasA = (targetKind & PK_HASACCS) != 0; q.hasT = (targetKind & PK_HASTHIRD) != 0;
   q.delWin = cfg_int(cfg, "DELTAWINDOW", 2); q.accWin = cfg_int(cfg, "ACCWINDOW", 2); q.thirdWin = cfg_int(cfg, "THIRDWINDOW", 2);
   q.nZeroMean = (add & PK_HASZEROM) ? base + (((targetKind & PK_HASZEROC) && !(targetKind & PK_HASNULLE)) ? 1 : 0) : 0;   /* HParm.c:1712-1715 */
   q.nullECol = ((targetKind & PK_HASNULLE) && nE) ? base : -1;
   q.v1Compat = cfg_bool(cfg, "V1COMPAT", 0); q.simpleDiffs = cfg_bool(cfg, "SIMPLEDIFFS", 0);
   ob->cols = htkamd_parm_quals_cols(&q);
   CHECK(htkamd_dev_malloc((void **)&ob->dX, sizeof(float) * (size_t)(F ? F : 1) * ob->cols));
   CHECK(htkamd_parm_qualify(dStat, ob->frameOff, count, &q, ob->dX, NULL));
   CHECK(htkamd_stream_sync(NULL));
   CHECK(htkamd_dev_free(dStat));
}

static void free_observations(obs_batch *ob) { if (ob->dX) htkamd_dev_free(ob->dX); free(ob->frameOff); memset(ob, 0, sizeof(*ob)); }

/* ---- option scanning in HTK's style: switches first ("-x", optionally followed by values), then positional arguments ---- */
typedef struct { int argc, at; char **argv; } args;
static int is_switch(const char *s) { return s[0] == '-' && s[1] && !isdigit((unsigned char)s[1]) && s[1] != '.'; }
static const char *next_switch(args *a) { return (a->at < a->argc && is_switch(a->argv[a->at])) ? a->argv[a->at++] + 1 : NULL; }
static const char *str_arg(args *a, const char *sw) { if (a->at >= a->argc) DIE("-%s: value expected", sw); return a->argv[a->at++]; }
static double flt_arg(args *a, const char *sw) { return atof(str_arg(a, sw)); }
static __attribute__((unused)) int has_num_arg(const args *a) { return a->at < a->argc && !is_switch(a->argv[a->at]) && (isdigit((unsigned char)a->argv[a->at][0]) || a->argv[a->at][0] == '.' || a->argv[a->at][0] == '-'); }

#endif
