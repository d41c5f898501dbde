'use strict';
/*
 * cascade_pack.js — pack / unpack a ccv BBF cascade object (the shape of `headtrackr.cascade`,
 * reference: /root/reference/src/cascade.js:19 — {count,width,height,stage_classifier:[{count,threshold,
 * feature:[{size,px,py,pz,nx,ny,nz}],alpha:[2*count]}]}) to/from the little-endian "HTCB" blob that the
 * C ABI (include/headtrackr_hip.h: ht_create) consumes.  The blob is a transport format only; the device
 * layout is built inside the library.
 *
 *   header  (32 B): 'HTCB', u32 version=1, u32 stages, u32 width, u32 height, u32 nfeat, u32 maxpts=8, u32 0
 *   stages  (16 B each): u32 count, u32 first_feature, f64 threshold
 *   features(72 B each): u8 size, 7 pad, i8 px[8], py[8], pz[8], nx[8], ny[8], nz[8], f64 alpha0, f64 alpha1
 *   unused point slots have z = -1 (and x = y = 0 in slots >= size).
 */
const MAXPTS = 8;
const HEADER = 32, STAGE_REC = 16, FEAT_REC = 72;

function packCascade(c) {
  const nst = c.stage_classifier.length;
  let nfeat = 0;
  for (let j = 0; j < nst; j++) nfeat += c.stage_classifier[j].count;
  const buf = Buffer.alloc(HEADER + nst * STAGE_REC + nfeat * FEAT_REC);
  buf.write('HTCB', 0, 'latin1');
  buf.writeUInt32LE(1, 4);
  buf.writeUInt32LE(nst, 8);
  buf.writeUInt32LE(c.width, 12);
  buf.writeUInt32LE(c.height, 16);
  buf.writeUInt32LE(nfeat, 20);
  buf.writeUInt32LE(MAXPTS, 24);
  let first = 0, fo = HEADER + nst * STAGE_REC;
  for (let j = 0; j < nst; j++) {
    const st = c.stage_classifier[j];
    const so = HEADER + j * STAGE_REC;
    buf.writeUInt32LE(st.count, so);
    buf.writeUInt32LE(first, so + 4);
    buf.writeDoubleLE(st.threshold, so + 8);
    for (let k = 0; k < st.count; k++, fo += FEAT_REC) {
      const f = st.feature[k];
      if (f.size > MAXPTS) throw new RangeError('cascade feature has more than ' + MAXPTS + ' points');
      buf.writeUInt8(f.size, fo);
      for (let q = 0; q < MAXPTS; q++) {
        /* slots with z < 0 are never read by the detector (ccv.js:198,208); their x/y are garbage in the
         * trained data (some do not even fit a byte), so they are normalised to 0 */
        const pu = q < f.size && f.pz[q] >= 0, nu = q < f.size && f.nz[q] >= 0;
        buf.writeInt8(pu ? f.px[q] : 0, fo + 8 + q);
        buf.writeInt8(pu ? f.py[q] : 0, fo + 16 + q);
        buf.writeInt8(pu ? f.pz[q] : -1, fo + 24 + q);
        buf.writeInt8(nu ? f.nx[q] : 0, fo + 32 + q);
        buf.writeInt8(nu ? f.ny[q] : 0, fo + 40 + q);
        buf.writeInt8(nu ? f.nz[q] : -1, fo + 48 + q);
      }
      buf.writeDoubleLE(st.alpha[2 * k], fo + 56);
      buf.writeDoubleLE(st.alpha[2 * k + 1], fo + 64);
    }
    first += st.count;
  }
  return buf;
}

function unpackCascade(buf) {
  if (buf.toString('latin1', 0, 4) !== 'HTCB' || buf.readUInt32LE(4) !== 1) throw new Error('not an HTCB v1 cascade blob');
  const nst = buf.readUInt32LE(8);
  const c = { count: nst, width: buf.readUInt32LE(12), height: buf.readUInt32LE(16), stage_classifier: [] };
  let fo = HEADER + nst * STAGE_REC;
  for (let j = 0; j < nst; j++) {
    const so = HEADER + j * STAGE_REC;
    const st = { count: buf.readUInt32LE(so), threshold: buf.readDoubleLE(so + 8), feature: [], alpha: [] };
    for (let k = 0; k < st.count; k++, fo += FEAT_REC) {
      const size = buf.readUInt8(fo);
      const f = { size: size, px: [], py: [], pz: [], nx: [], ny: [], nz: [] };
      for (let q = 0; q < size; q++) {
        f.px.push(buf.readInt8(fo + 8 + q)); f.py.push(buf.readInt8(fo + 16 + q)); f.pz.push(buf.readInt8(fo + 24 + q));
        f.nx.push(buf.readInt8(fo + 32 + q)); f.ny.push(buf.readInt8(fo + 40 + q)); f.nz.push(buf.readInt8(fo + 48 + q));
      }
      st.feature.push(f);
      st.alpha.push(buf.readDoubleLE(fo + 56), buf.readDoubleLE(fo + 64));
    }
    c.stage_classifier.push(st);
  }
  return c;
}

module.exports = { packCascade: packCascade, unpackCascade: unpackCascade, MAXPTS: MAXPTS };
