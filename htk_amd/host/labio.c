/* labio.c -- HTK label files and master label files (MLF), the transcription inputs of HERest -L/-I and HVite -a.
 *
 * Replaces LoadHTKLabels (HLabel.c:748-840: "[start [end]] name [score] {aux}" per line, times in 100 ns units,
 * a line of "///" separates alternatives -- only the first is kept, as HERest/HVite -a use it) and the immediate-definition
 * part of LoadMasterFile / the MLF search (HLabel.c:1410-1560: "#!MLF!#", then per entry a quoted pattern line and label
 * lines up to a single "."; patterns match with '*' and '?' like MaskMatch).  Sub-directory redirection ("->" / "=>")
 * is not supported.
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

struct htkamd_labels {
   int n;
   char **name;
   long long *start, *end;       /* -1 when absent */
   float *score;                 /* 0 when absent */
};

struct htkamd_mlf {
   int n;
   char **pattern;
   struct htkamd_labels **lab;
};

static int is_number(const char *t)
{
   char *e;
   if (!*t) return 0;
   strtod(t, &e);
   return *e == 0;
}

/* one label line -> appended to L; returns 0, or 1 for the alternative separator */
static int add_line(struct htkamd_labels *L, int *cap, char *line)
{
   char *tok[8]; int nt = 0;
   for (char *p = strtok(line, " \t\r\n"); p && nt < 8; p = strtok(NULL, " \t\r\n")) tok[nt++] = p;
   if (nt == 0) return 0;
   if (!strcmp(tok[0], "///")) return 1;
   long long st = -1, en = -1; float sc = 0.0f; int k = 0;
   if (nt >= 2 && is_number(tok[0]) && (isdigit((unsigned char)tok[0][0]) || tok[0][0] == '-')) {
      st = atoll(tok[k++]);
      if (nt >= 3 && is_number(tok[1])) en = atoll(tok[k++]);
   }
   char *name = tok[k++];
   size_t L0 = strlen(name);
   if (L0 >= 2 && (name[0] == '"' || name[0] == '\'') && name[L0 - 1] == name[0]) { name[L0 - 1] = 0; name++; }
   if (k < nt && is_number(tok[k])) sc = strtof(tok[k], NULL);
   if (L->n + 1 > *cap) {
      *cap = *cap * 2 + 32;
      L->name = (char **)realloc(L->name, sizeof(char *) * (size_t)*cap);
      L->start = (long long *)realloc(L->start, sizeof(long long) * (size_t)*cap);
      L->end = (long long *)realloc(L->end, sizeof(long long) * (size_t)*cap);
      L->score = (float *)realloc(L->score, sizeof(float) * (size_t)*cap);
   }
   L->name[L->n] = strdup(name); L->start[L->n] = st; L->end[L->n] = en; L->score[L->n] = sc; L->n++;
   return 0;
}

void htkamd_labels_free(struct htkamd_labels *L)
{
   if (!L) return;
   for (int i = 0; i < L->n; i++) free(L->name[i]);
   free(L->name); free(L->start); free(L->end); free(L->score); free(L);
}

int htkamd_labels_read(const char *path, struct htkamd_labels **out)
{
   if (!path || !out) { htkamd_set_error("labels_read: NULL argument"); return HTKAMD_EINVAL; }
   FILE *f = fopen(path, "r");
   if (!f) { htkamd_set_error("labels_read: cannot open %s", path); return HTKAMD_EIO; }
   struct htkamd_labels *L = (struct htkamd_labels *)calloc(1, sizeof(*L));
   int cap = 0, alt = 0;
   char line[2048];
   while (fgets(line, sizeof(line), f)) {
      if (alt) continue;
      if (add_line(L, &cap, line)) alt = 1;
   }
   fclose(f);
   *out = L;
   return HTKAMD_OK;
}

int htkamd_labels_count(const struct htkamd_labels *L) { return L ? L->n : 0; }
const char *htkamd_labels_name(const struct htkamd_labels *L, int i) { return (L && i >= 0 && i < L->n) ? L->name[i] : NULL; }
long long htkamd_labels_start(const struct htkamd_labels *L, int i) { return (L && i >= 0 && i < L->n) ? L->start[i] : -1; }
long long htkamd_labels_end(const struct htkamd_labels *L, int i) { return (L && i >= 0 && i < L->n) ? L->end[i] : -1; }
float htkamd_labels_score(const struct htkamd_labels *L, int i) { return (L && i >= 0 && i < L->n) ? L->score[i] : 0.0f; }

/* MaskMatch-style wildcard match ('*' any run, '?' one character) */
static int wild(const char *p, const char *s)
{
   if (*p == 0) return *s == 0;
   if (*p == '*') { for (;; s++) { if (wild(p + 1, s)) return 1; if (*s == 0) return 0; } }
   if (*s == 0) return 0;
   if (*p == '?' || *p == *s) return wild(p + 1, s + 1);
   return 0;
}

void htkamd_mlf_free(struct htkamd_mlf *m)
{
   if (!m) return;
   for (int i = 0; i < m->n; i++) { free(m->pattern[i]); htkamd_labels_free(m->lab[i]); }
   free(m->pattern); free(m->lab); free(m);
}

int htkamd_mlf_read(const char *path, struct htkamd_mlf **out)
{
   if (!path || !out) { htkamd_set_error("mlf_read: NULL argument"); return HTKAMD_EINVAL; }
   FILE *f = fopen(path, "r");
   if (!f) { htkamd_set_error("mlf_read: cannot open %s", path); return HTKAMD_EIO; }
   char line[2048];
   if (!fgets(line, sizeof(line), f) || strncmp(line, "#!MLF!#", 7)) { fclose(f); htkamd_set_error("mlf_read: %s has no #!MLF!# header", path); return HTKAMD_EMODEL; }
   struct htkamd_mlf *m = (struct htkamd_mlf *)calloc(1, sizeof(*m));
   int capM = 0;
   while (fgets(line, sizeof(line), f)) {
      char *p = line;
      while (isspace((unsigned char)*p)) p++;
      if (!*p) continue;
      char *e = p + strlen(p);
      while (e > p && isspace((unsigned char)e[-1])) *--e = 0;
      if (strstr(p, "->") || strstr(p, "=>")) { fclose(f); htkamd_mlf_free(m); htkamd_set_error("mlf_read: %s: sub-directory entries are not supported", path); return HTKAMD_EMODEL; }
      if ((*p == '"' || *p == '\'') && e > p + 1 && e[-1] == *p) { e[-1] = 0; p++; }
      if (m->n + 1 > capM) { capM = capM * 2 + 32; m->pattern = (char **)realloc(m->pattern, sizeof(char *) * (size_t)capM); m->lab = (struct htkamd_labels **)realloc(m->lab, sizeof(void *) * (size_t)capM); }
      char *pattern = strdup(p);                     /* `line` is reused for the label lines below */
      struct htkamd_labels *L = (struct htkamd_labels *)calloc(1, sizeof(*L));
      int cap = 0, alt = 0, closed = 0;
      while (fgets(line, sizeof(line), f)) {
         char *q = line;
         while (isspace((unsigned char)*q)) q++;
         if (q[0] == '.' && (q[1] == 0 || isspace((unsigned char)q[1]))) { closed = 1; break; }
         if (alt) continue;
         if (add_line(L, &cap, line)) alt = 1;
      }
      (void)closed;                                  /* a last entry may end at EOF without '.' */
      m->pattern[m->n] = pattern; m->lab[m->n] = L; m->n++;
   }
   fclose(f);
   *out = m;
   return HTKAMD_OK;
}

/* Labels of `labFile` (e.g. "lab/u00001.lab"; callers derive it from the data file name as HERest does: base name + ".lab"):
   first entry whose pattern matches; NULL if none. */
const struct htkamd_labels *htkamd_mlf_find(const struct htkamd_mlf *m, const char *labFile)
{
   if (!m || !labFile) return NULL;
   for (int i = 0; i < m->n; i++) if (wild(m->pattern[i], labFile)) return m->lab[i];
   return NULL;
}

/* ---- script files (-S): the list of data files of a tool run ------------------------------------------------------------
 * ScriptWord (HShell.c:661-688): words separated by white space, or enclosed in single / double quotes (no escapes);
 * RegisterExtFileName (HShell.c:86-140): a word may be an extended file name  logical=physical[start,end]  (either part optional):
 * the data are read from `physical`, frames start..end inclusive, and the file is known (label lookup) as `logical`. */
struct htkamd_scp {
   int n;
   char **logical, **physical;
   long *start, *end;            /* -1 = whole file */
};

void htkamd_scp_free(struct htkamd_scp *s)
{
   if (!s) return;
   for (int i = 0; i < s->n; i++) { free(s->logical[i]); free(s->physical[i]); }
   free(s->logical); free(s->physical); free(s->start); free(s->end); free(s);
}

int htkamd_scp_read(const char *path, struct htkamd_scp **out)
{
   if (!path || !out) { htkamd_set_error("scp_read: NULL argument"); return HTKAMD_EINVAL; }
   FILE *f = fopen(path, "r");
   if (!f) { htkamd_set_error("scp_read: cannot open script file %s", path); return HTKAMD_EIO; }
   struct htkamd_scp *s = (struct htkamd_scp *)calloc(1, sizeof(*s));
   int cap = 0, ch;
   char buf[4096];
   for (;;) {
      int i = 0;
      do ch = fgetc(f); while (ch != EOF && isspace(ch));
      if (ch == EOF) break;
      if (ch == '\'' || ch == '"') {
         const int q = ch;
         while ((ch = fgetc(f)) != q && ch != EOF) if (i < (int)sizeof(buf) - 1) buf[i++] = (char)ch;
         if (ch == EOF) { fclose(f); htkamd_scp_free(s); htkamd_set_error("scp_read: %s: closing quote missing", path); return HTKAMD_EINVAL; }
      } else {
         do { if (i < (int)sizeof(buf) - 1) buf[i++] = (char)ch; ch = fgetc(f); } while (ch != EOF && !isspace(ch));
      }
      buf[i] = 0;
      if (s->n + 1 > cap) {
         cap = cap * 2 + 64;
         s->logical = (char **)realloc(s->logical, sizeof(char *) * (size_t)cap); s->physical = (char **)realloc(s->physical, sizeof(char *) * (size_t)cap);
         s->start = (long *)realloc(s->start, sizeof(long) * (size_t)cap); s->end = (long *)realloc(s->end, sizeof(long) * (size_t)cap);
      }
      long st = -1, en = -1;
      char *eq = strchr(buf, '='), *lb = strchr(buf, '[');
      if (lb) {
         char *co = strchr(buf, ','), *rb = strchr(buf, ']');
         if (!co || !rb) { fclose(f); htkamd_scp_free(s); htkamd_set_error("scp_read: %s: bad index spec in %s", path, buf); return HTKAMD_EINVAL; }
         *rb = 0; en = atol(co + 1);
         *co = 0; st = atol(lb + 1);
         *lb = 0;
      }
      if (eq) { *eq = 0; s->physical[s->n] = strdup(eq + 1); s->logical[s->n] = strdup(buf); }
      else { s->physical[s->n] = strdup(buf); s->logical[s->n] = strdup(buf); }
      s->start[s->n] = st; s->end[s->n] = en; s->n++;
   }
   fclose(f);
   *out = s;
   return HTKAMD_OK;
}

int htkamd_scp_count(const struct htkamd_scp *s) { return s ? s->n : 0; }
const char *htkamd_scp_logical(const struct htkamd_scp *s, int i) { return (s && i >= 0 && i < s->n) ? s->logical[i] : NULL; }
const char *htkamd_scp_physical(const struct htkamd_scp *s, int i) { return (s && i >= 0 && i < s->n) ? s->physical[i] : NULL; }
long htkamd_scp_start(const struct htkamd_scp *s, int i) { return (s && i >= 0 && i < s->n) ? s->start[i] : -1; }
long htkamd_scp_end(const struct htkamd_scp *s, int i) { return (s && i >= 0 && i < s->n) ? s->end[i] : -1; }

/* ---- label output: a transcription being built, FormatTranscription and SaveHTKLabels ---------------------------------------
 * One label list (one level) with up to two auxiliary labels per entry, as TranscriptionFromLattice builds it for HVite
 * (HRec.c:2176-2360: word level = no auxiliary labels; -m = [word]; -f -m = [model, word]); htkamd_trans_format applies HVite's
 * -o flags the way FormatTranscription does (HRec.c:2368-2472); htkamd_trans_write prints like SaveHTKLabels (HLabel.c:1481-1525:
 * times "%.0f" in 100 ns units when present, a score column only if ANY label of the list has a non-zero score in it). */
#include <math.h>

typedef struct { double start, end; char *name; float score; char *aux[2]; float auxScore[2]; } trans_lab;
struct htkamd_trans { int n, cap, maxAux; trans_lab *lab; struct htkamd_trans *next; };     /* next: the next alternative (label list) */

int htkamd_trans_create(int maxAux, struct htkamd_trans **out)
{
   if (!out || maxAux < 0 || maxAux > 2) { htkamd_set_error("trans_create: bad argument"); return HTKAMD_EINVAL; }
   struct htkamd_trans *t = (struct htkamd_trans *)calloc(1, sizeof(*t));
   t->maxAux = maxAux;
   *out = t;
   return HTKAMD_OK;
}

void htkamd_trans_free(struct htkamd_trans *t)
{
   while (t) {
      struct htkamd_trans *nx = t->next;
      for (int i = 0; i < t->n; i++) { free(t->lab[i].name); free(t->lab[i].aux[0]); free(t->lab[i].aux[1]); }
      free(t->lab); free(t);
      t = nx;
   }
}

/* N-best output: `alt` becomes the last alternative of `t` (a Transcription with several label lists, AddLabelList HLabel.c;
   written one after the other with a line of "///" between them, SaveHTKLabels :1522).  `t` owns it from here on. */
int htkamd_trans_append_alternative(struct htkamd_trans *t, struct htkamd_trans *alt)
{
   if (!t || !alt || t == alt) { htkamd_set_error("trans_append_alternative: bad argument"); return HTKAMD_EINVAL; }
   while (t->next) t = t->next;
   t->next = alt;
   return HTKAMD_OK;
}

/* start / end in 100 ns units (-1 = absent); aux1 / aux2 may be NULL */
int htkamd_trans_add(struct htkamd_trans *t, double start, double end, const char *name, float score,
                     const char *aux1, float aux1Score, const char *aux2, float aux2Score)
{
   if (!t || !name) { htkamd_set_error("trans_add: NULL argument"); return HTKAMD_EINVAL; }
   if (t->n + 1 > t->cap) { t->cap = t->cap * 2 + 32; t->lab = (trans_lab *)realloc(t->lab, sizeof(trans_lab) * (size_t)t->cap); }
   trans_lab *l = &t->lab[t->n++];
   l->start = start; l->end = end; l->name = strdup(name); l->score = score;
   l->aux[0] = (aux1 && t->maxAux >= 1) ? strdup(aux1) : NULL; l->auxScore[0] = aux1Score;
   l->aux[1] = (aux2 && t->maxAux >= 2) ? strdup(aux2) : NULL; l->auxScore[1] = aux2Score;
   return HTKAMD_OK;
}

static void tri_strip_inplace(char *s)                       /* TriStrip (HLabel/HUtil): a-b+c -> b */
{
   char *p = strchr(s, '-');
   if (p) memmove(s, p + 1, strlen(p + 1) + 1);
   if ((p = strrchr(s, '+')) != NULL) *p = 0;
}

/* flags: the letters of HVite -o / HTKAMD_OUT_* bits.  frameDur in 100 ns units; states / models = the -f / -m switches. */
int htkamd_trans_format(struct htkamd_trans *t, double frameDur, int states, int models, int flags)
{
   if (!t || frameDur <= 0) { htkamd_set_error("trans_format: bad argument"); return HTKAMD_EINVAL; }
   if (t->next) { const int rc = htkamd_trans_format(t->next, frameDur, states, models, flags); if (rc) return rc; }
   if (flags & HTKAMD_OUT_NOSCORES)
      for (int i = 0; i < t->n; i++) { t->lab[i].score = 0.0f; t->lab[i].auxScore[0] = t->lab[i].auxScore[1] = 0.0f; }
   if (flags & HTKAMD_OUT_TRISTRIP)
      for (int i = 0; i < t->n; i++) {
         trans_lab *l = &t->lab[i];
         if (states && !models) {                             /* "model[state]" names: keep the tail */
            char tail[64] = "", *p = strrchr(l->name, '[');
            if (p) { snprintf(tail, sizeof(tail), "%s", p); *p = 0; }
            tri_strip_inplace(l->name);
            char *nn = (char *)malloc(strlen(l->name) + strlen(tail) + 1);
            strcpy(nn, l->name); strcat(nn, tail); free(l->name); l->name = nn;
         } else if (!states && models) tri_strip_inplace(l->name);
         else if (states && models && l->aux[0]) tri_strip_inplace(l->aux[0]);
      }
   if (flags & HTKAMD_OUT_NORMSCORES)
      for (int i = 0; i < t->n; i++) {
         trans_lab *l = &t->lab[i];
         int frames = (int)floor((l->end - l->start) / frameDur + 0.4);
         l->score = frames == 0 ? 0.0f : l->score / frames;
         if (states && models && t->maxAux > 0 && l->aux[0]) {          /* the model spans the labels up to the next model label */
            double end = l->end;
            for (int k = i + 1; k < t->n && !t->lab[k].aux[0]; k++) end = t->lab[k].end;
            frames = (int)floor((end - l->start) / frameDur + 0.4);
            l->auxScore[0] = frames == 0 ? 0.0f : l->auxScore[0] / frames;
         }
      }
   if (flags & HTKAMD_OUT_NOTIMES)
      for (int i = 0; i < t->n; i++) t->lab[i].start = t->lab[i].end = -1.0;
   if (flags & HTKAMD_OUT_CENTRE)
      for (int i = 0; i < t->n; i++) { t->lab[i].start += frameDur / 2; t->lab[i].end -= frameDur / 2; }
   if ((flags & HTKAMD_OUT_NOWORDS) && t->maxAux > 0)
      for (int i = 0; i < t->n; i++) { free(t->lab[i].aux[t->maxAux - 1]); t->lab[i].aux[t->maxAux - 1] = NULL; }
   if ((flags & HTKAMD_OUT_NOMODELS) && models && states && t->maxAux == 2)
      for (int i = 0; i < t->n; i++) {
         trans_lab *l = &t->lab[i];
         free(l->aux[0]); l->aux[0] = l->aux[1]; l->auxScore[0] = l->auxScore[1]; l->aux[1] = NULL;
      }
   return HTKAMD_OK;
}

static void write_name(FILE *f, const char *s)               /* WriteString with q = 0 (HShell.c:1268): quote only names that start with one */
{
   int q = 0;
   if (s[0] == '"') q = '\''; else if (s[0] == '\'') q = '"';
   if (q) fputc(q, f);
   for (const unsigned char *p = (const unsigned char *)s; *p; p++) {
      if (*p == '\\' || (q && *p == q)) { fputc('\\', f); fputc(*p, f); }
      else if (*p >= 32 && *p < 127) fputc(*p, f);
      else fprintf(f, "\\%c%c%c", ((*p / 64) % 8) + '0', ((*p / 8) % 8) + '0', (*p % 8) + '0');
   }
   if (q) fputc(q, f);
}

static void trans_print_one(FILE *f, const struct htkamd_trans *t);
static void trans_print(FILE *f, const struct htkamd_trans *t)
{
   for (; t; t = t->next) { trans_print_one(f, t); if (t->next) fprintf(f, "///\n"); }
}
static void trans_print_one(FILE *f, const struct htkamd_trans *t)
{
   int has[3] = {0, 0, 0};
   for (int i = 0; i < t->n; i++) {
      if (t->lab[i].score != 0.0f) has[0] = 1;
      for (int j = 0; j < t->maxAux; j++) if (t->lab[i].auxScore[j] != 0.0f) has[j + 1] = 1;
   }
   for (int i = 0; i < t->n; i++) {
      const trans_lab *l = &t->lab[i];
      if (l->start >= 0.0) {
         fprintf(f, "%.0f ", l->start);
         if (l->end >= 0.0) fprintf(f, "%.0f ", l->end);
      }
      write_name(f, l->name);
      if (has[0]) fprintf(f, " %f", l->score);
      for (int j = 0; j < t->maxAux; j++)
         if (l->aux[j]) { fputc(' ', f); write_name(f, l->aux[j]); if (has[j + 1]) fprintf(f, " %f", l->auxScore[j]); }
      fprintf(f, "\n");
   }
}

int htkamd_trans_write(const struct htkamd_trans *t, const char *path)
{
   if (!t || !path) { htkamd_set_error("trans_write: NULL argument"); return HTKAMD_EINVAL; }
   FILE *f = fopen(path, "w");
   if (!f) { htkamd_set_error("trans_write: cannot create %s", path); return HTKAMD_EIO; }
   trans_print(f, t);
   fclose(f);
   return HTKAMD_OK;
}

/* master label file output (HVite -i): "#!MLF!#", then per transcription its quoted name, the labels and "." */
struct htkamd_mlf_out { FILE *f; };
int htkamd_mlf_out_open(const char *path, struct htkamd_mlf_out **out)
{
   if (!path || !out) { htkamd_set_error("mlf_out_open: NULL argument"); return HTKAMD_EINVAL; }
   FILE *f = fopen(path, "w");
   if (!f) { htkamd_set_error("mlf_out_open: cannot create %s", path); return HTKAMD_EIO; }
   fprintf(f, "#!MLF!#\n");
   struct htkamd_mlf_out *o = (struct htkamd_mlf_out *)calloc(1, sizeof(*o));
   o->f = f; *out = o;
   return HTKAMD_OK;
}
int htkamd_mlf_out_add(struct htkamd_mlf_out *o, const char *labFile, const struct htkamd_trans *t)
{
   if (!o || !labFile || !t) { htkamd_set_error("mlf_out_add: NULL argument"); return HTKAMD_EINVAL; }
   fprintf(o->f, "\"%s\"\n", labFile);
   trans_print(o->f, t);
   fprintf(o->f, ".\n");
   return HTKAMD_OK;
}
void htkamd_mlf_out_close(struct htkamd_mlf_out *o) { if (o) { fclose(o->f); free(o); } }
