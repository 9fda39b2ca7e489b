// Vaa3D shell around libpnr_hip.so.  NOT built in this repository's tests: it needs Qt4 and a Vaa3D source tree
// (v3d_interface.h), neither of which is available where the library is developed.  It exists so that
// `vaa3d -x Advantra -f advantra_func -i <img> -p <11 parameters>` keeps working with the hot path on the GPU.
// The interface is the one every Vaa3D plugin implements; the reference's is /root/reference/pnr-vaa3d/Advantra_plugin.h:11-24.
#pragma once
#include <QtGui>
#include <v3d_interface.h>

class Advantra : public QObject, public V3DPluginInterface2_1 {
    Q_OBJECT
    Q_INTERFACES(V3DPluginInterface2_1);

  public:
    float getPluginVersion() const { return 1.1f; }
    QStringList menulist() const;
    void domenu(const QString &menu_name, V3DPluginCallback2 &callback, QWidget *parent);
    QStringList funclist() const;
    bool dofunc(const QString &func_name, const V3DPluginArgList &input, V3DPluginArgList &output, V3DPluginCallback2 &callback,
                QWidget *parent);
};
