/* orc_ilist.h -- HRec's list of network-node instances, restated (TEST INFRASTRUCTURE; shared by orc_decode.c and orc_decode_n.c).
 *
 * The reference keeps one NetInst per active network node on a doubly linked list between the sentinels pri->head and pri->tail
 * (HRec.c:197-205).  Three operations change it:
 *   AttachInst   (HRec.c:1200-1268)  a new instance is appended before the tail, marked out-of-order, and ReOrderList runs on its node
 *   MoveToRecent (HRec.c:1123-1150)  an instance is unlinked and appended again (pxd cleared, ooo set); if it is the instance pass 2
 *                                    is standing on, the walk continues from its predecessor (pri->nxtInst)
 *   ReOrderList  (HRec.c:1152-1170)  for a node whose instance is out of order: every LIVE successor over a zero-time link (the tr0
 *                                    links come first in a node's link array, ExpandWordNet HNet.c:3632-3645) is moved to the tail in
 *                                    link order, then ReOrderList recurses into each of them in link order
 *   DetachInst   (HRec.c:1270-1301)  the instance is unlinked
 * so that an instance always lies behind every live instance that can send it a token within the frame, and pass 2 -- one walk from
 * the head, while the list changes under it -- steps every node after its senders.  A node has at most one instance, so the node's
 * index stands for the instance here: link[] / knil[] over nNodes + 2 entries, HEAD = nNodes, TAIL = nNodes + 1.
 */
#ifndef ORC_ILIST_H
#define ORC_ILIST_H

typedef struct {
   int nNodes;
   const int *linkOff, *linkDest;
   int *link, *knil;            /* [nNodes + 2] */
   char *att, *ooo, *tr0;       /* [nNodes]: has an instance; inst->ooo; node_tr0(node) */
   int nxtInst;                 /* pri->nxtInst */
} orc_ilist;

static void orc_ilist_append(orc_ilist *L, int n)
{
   const int TAIL = L->nNodes + 1;
   L->link[n] = TAIL; L->knil[n] = L->knil[TAIL];
   L->knil[TAIL] = n; L->link[L->knil[n]] = n;
}

static void orc_ilist_unlink(orc_ilist *L, int n)
{
   L->knil[L->link[n]] = L->knil[n];
   L->link[L->knil[n]] = L->link[n];
}

static void orc_ilist_move_to_recent(orc_ilist *L, int n)
{
   if (n == L->nxtInst) L->nxtInst = L->knil[n];
   orc_ilist_unlink(L, n);
   orc_ilist_append(L, n);
   L->ooo[n] = 1;
}

static void orc_ilist_reorder(orc_ilist *L, int n)
{
   int k;
   if (!L->att[n] || !L->ooo[n]) return;
   L->ooo[n] = 0;
   for (k = L->linkOff[n]; k < L->linkOff[n + 1]; k++) {
      const int dst = L->linkDest[k];
      if (!L->tr0[dst]) break;
      if (L->att[dst]) orc_ilist_move_to_recent(L, dst);
   }
   for (k = L->linkOff[n]; k < L->linkOff[n + 1]; k++) {
      const int dst = L->linkDest[k];
      if (!L->tr0[dst]) break;
      if (L->att[dst]) orc_ilist_reorder(L, dst);
   }
}

static void orc_ilist_attach(orc_ilist *L, int n)
{
   L->att[n] = 1;
   orc_ilist_append(L, n);
   L->ooo[n] = 1;
   orc_ilist_reorder(L, n);
}

static void orc_ilist_detach(orc_ilist *L, int n)
{
   orc_ilist_unlink(L, n);
   L->att[n] = 0;
}

#endif
